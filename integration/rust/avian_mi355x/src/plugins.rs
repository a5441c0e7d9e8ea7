//! The replacement plugins.  Every system is added into the system set of the reference system it replaces, so the rest of Avian
//! (narrow phase, sleeping, spatial queries, transform sync, user systems ordered against Avian's sets) keeps its ordering.

use crate::{staging::Staging, world::{default_config, Mi355xWorld}};
use avian3d::{
    collision::{broad_phase::BroadPhaseSystems, narrow_phase::NarrowPhaseConfig},
    dynamics::{
        integrator::IntegrationSystems,
        rigid_body::forces::AccumulatedLocalAcceleration,
        solver::{
            constraint_graph::ConstraintGraph,
            schedule::{SolverSystems, SubstepCount},
            solver_body::SolverBody,
            SolverConfig, SolverDiagnostics,
        },
    },
    collision::CollisionDiagnostics,
    prelude::*,
};
use avian_mi355x_sys as ffi;
use bevy::ecs::system::{StaticSystemParam, SystemParamItem};
use bevy::prelude::*;
use core::time::Duration;

/// How much of the step stays on the device.
#[derive(Clone, Copy, Default, PartialEq, Eq)]
pub enum Mi355xMode {
    /// The default: `ClosedLoop` whenever the scene allows it -- every collider on a rigid body the staging knows (Ball / Cuboid in device kernels, every other
    /// shape through the host-shape callbacks of host_shapes.rs; `ActiveCollisionHooks` colliders through the hook callbacks of hooks.rs) -- and `HostNarrowPhase` otherwise, decided per step (`Mi355xSettings::effective_mode`).  The host-manifold
    /// flow moves ~160 MB over PCIe per cfg2 step (5 ms) where the closed loop moves three counter blocks: it must not be what a user
    /// gets without asking.
    #[default]
    Auto,
    /// Avian's own (parry) narrow phase keeps running on the host; its manifolds are uploaded every step (`avn_manifolds_upload`),
    /// the substep loop runs device-resident.  Works with every collider shape; pays the manifold upload over PCIe.
    HostNarrowPhase,
    /// Contact rows live in HBM (Ball / Cuboid manifolds from device kernels, other shapes' from `Collider::contact_manifolds` through the callback), `NarrowPhase::update_contacts`, the status-change loop, the
    /// `ConstraintGraph` and the `IdPool` run on the device (`avn_pipeline_enable(1)`); per step only new pairs and body state cross
    /// the bus.  Collision events / `CollidingEntities` are rebuilt from `avn_pipeline_new_pair_ids_get` + `avn_contact_changes_get` by
    /// `closed_loop::gpu_closed_loop_events`; sleeping is the library's own island manager (`avn_sleeping_enable`), mirrored into `Sleeping` /
    /// `SleepTimer` by `closed_loop::gpu_closed_loop_sleeping`.
    /// `CollisionHooks` (src/collision/hooks.rs:147-186) are called back from inside `avn_step` (hooks.rs, `avn_collision_hooks_set`): `filter_pairs` before a
    /// candidate pair gets its `ContactId`, `modify_contacts` between `prune_points` and `match_contacts` -- only the pairs of `ActiveCollisionHooks` colliders
    /// cross the bus (until round 6 one such collider sent the whole world to `HostNarrowPhase` mode).
    ClosedLoop,
}

/// `H`: the application's `CollisionHooks` system parameter, exactly as for `PhysicsPlugins::with_collision_hooks::<H>()` (`()` = no hooks).
pub struct Mi355xPhysicsPlugin<H: CollisionHooks + 'static = ()> {
    pub mode: Mi355xMode,
    pub device: i32,
    _hooks: core::marker::PhantomData<H>,
}
impl<H: CollisionHooks + 'static> Default for Mi355xPhysicsPlugin<H> {
    fn default() -> Self { Self { mode: Mi355xMode::default(), device: 0, _hooks: core::marker::PhantomData } }
}

#[derive(Resource, Default)]
pub struct Mi355xStaging(pub Staging);
#[derive(Resource)]
struct Mi355xSettings { mode: Mi355xMode, warned_hooks: bool }
impl Mi355xSettings {
    /// ClosedLoop unless a collider sits on something the staging does not know as a rigid body (colliders with `ActiveCollisionHooks` stay: their pairs go through
    /// `filter_pairs` / `modify_contacts` by callback, hooks.rs); the switch is per step -- leaving / entering the device loop is avn_pipeline_enable(0 / 1),
    /// which drops / rebuilds the device contact rows (one step without warm starting, like any ContactGraph rebuild).
    fn effective_mode(&mut self, st: &Staging) -> Mi355xMode {
        if self.mode == Mi355xMode::HostNarrowPhase { return Mi355xMode::HostNarrowPhase; }
        if st.colliders_unsupported != 0 {
            if !self.warned_hooks && self.mode == Mi355xMode::ClosedLoop {
                bevy::log::warn!("avian_mi355x: {} collider(s) are not on a known rigid body or have a composite shape (trimesh, heightfield, compound: several manifolds per pair): running in HostNarrowPhase mode", st.colliders_unsupported);
                self.warned_hooks = true;
            }
            return Mi355xMode::HostNarrowPhase;
        }
        Mi355xMode::ClosedLoop
    }
}

impl<H: CollisionHooks + 'static> Plugin for Mi355xPhysicsPlugin<H>
where
    for<'w, 's> SystemParamItem<'w, 's, H>: CollisionHooks,
{
    fn build(&self, app: &mut App) {
        let mut config = default_config();
        config.device = self.device;
        let world = match Mi355xWorld::new(config) {
            Ok(w) => w,
            // no gfx950 device: there is no CPU fallback in the library -- the application keeps the stock plugins
            Err(e) => panic!("{e}: do not disable Avian's BroadPhasePlugin / IntegratorPlugin / SolverPlugin / XpbdSolverPlugin on a host without an MI355X"),
        };
        app.insert_resource(world).init_resource::<Mi355xStaging>().insert_resource(Mi355xSettings { mode: self.mode, warned_hooks: false });
        app.init_resource::<crate::joints::JointStaging>().init_resource::<crate::closed_loop::ContactMirror>().init_resource::<crate::closed_loop::ClosedLoopSleeping>();
        app.init_resource::<crate::host_shapes::HostShapeTable>();   // colliders that are not Ball / Cuboid: answered through avn_host_shapes_set (registered in gpu_upload_bodies)
        app.init_resource::<SolverDiagnostics>().init_resource::<CollisionDiagnostics>();

        // the same sets as the plugins being replaced: src/collision/broad_phase.rs:51-74, src/dynamics/solver/plugin.rs:103-150,
        // src/dynamics/integrator/mod.rs:52-71
        app.add_systems(
            PhysicsSchedule,
            (
                (sync_config, gpu_upload_bodies, gpu_broad_phase).chain().in_set(BroadPhaseSystems::CollectCollisions),
                // XpbdSolverPlugin's PrepareJoints: the step's joint set of all five types, with JointDamping and JointCollisionDisabled
                crate::joints::gpu_upload_joints.in_set(SolverSystems::PrepareJoints),
                gpu_upload_constraints.in_set(SolverSystems::PrepareContactConstraints),   // (HostNarrowPhase steps only: checked inside)
                gpu_solver::<H>.in_set(SolverSystems::Substep),   // prepare + ALL substeps + restitution, device resident (AVN_SYS_SOLVER); in the closed loop the user's CollisionHooks are called back from inside it
                gpu_download.in_set(SolverSystems::StoreContactImpulses),
                crate::joints::gpu_download_joints.in_set(SolverSystems::Writeback),   // writeback_joint_forces::<T> of all five types (xpbd/plugin.rs:242-260)
                // closed loop only (each returns at once otherwise): the status loop's events, then the library's sleeping verdict into the ECS
                (crate::closed_loop::gpu_closed_loop_events, crate::closed_loop::gpu_closed_loop_sleeping).chain().after(SolverSystems::StoreContactImpulses),
                crate::closed_loop::gpu_closed_loop_wake_on_changed.before(BroadPhaseSystems::CollectCollisions),
                gpu_diagnostics.after(SolverSystems::StoreContactImpulses),
                // replaces update_sleeping_states (src/dynamics/solver/islands/sleeping.rs:71-83 chains it before sleep_islands): the timers and
                // the per-island decision come from the device, the commands that act on it stay Avian's own
                // (HostNarrowPhase steps only: in the closed loop Avian's PhysicsIslands sees no contact -- its narrow phase walks an empty pair list -- and
                //  the library's own island manager decides, see closed_loop.rs)
                gpu_sleeping.in_set(PhysicsStepSystems::Sleeping).run_if(resource_exists::<PhysicsIslands>),
            ),
        );
        // IntegratorPlugin's own bookkeeping set stays meaningful for user systems ordered against it
        app.configure_sets(PhysicsSchedule, IntegrationSystems::UpdateVelocityIncrements.before(SolverSystems::Substep));
    }
}

/// `SubstepCount`, `Gravity`, `SolverConfig`, `PhysicsLengthUnit`, `NarrowPhaseConfig`, the step's delta -> `avn_config_set` when any changed.
fn sync_config(
    mut w: ResMut<Mi355xWorld>, substeps: Res<SubstepCount>, gravity: Res<Gravity>, solver: Res<SolverConfig>, length_unit: Res<PhysicsLengthUnit>,
    narrow: Res<NarrowPhaseConfig>, time: Res<Time<Physics>>,
) {
    let mut c = w.config;
    c.substeps = substeps.0;
    c.dt_ns = time.delta().as_nanos() as u64;
    c.gravity = [gravity.0.x as f64, gravity.0.y as f64, gravity.0.z as f64];
    c.length_unit = length_unit.0 as f64;
    c.contact_damping_ratio = solver.contact_damping_ratio as f64;
    c.contact_frequency_factor = solver.contact_frequency_factor as f64;
    c.max_overlap_solve_speed = solver.max_overlap_solve_speed as f64;
    c.warm_start_coefficient = solver.warm_start_coefficient as f64;
    c.restitution_threshold = solver.restitution_threshold as f64;
    c.restitution_iterations = solver.restitution_iterations as u32;
    c.match_contacts = narrow.match_contacts as u32;
    c.default_speculative_margin = narrow.default_speculative_margin as f64;
    c.contact_tolerance = narrow.contact_tolerance as f64;
    if c.dt_ns != 0 && (c.substeps, c.dt_ns, c.gravity, c.length_unit) != (w.config.substeps, w.config.dt_ns, w.config.gravity, w.config.length_unit)
        || solver.is_changed() || narrow.is_changed()
    {
        w.set_config(c);
    }
}

/// ECS -> device: what `prepare_solver_bodies` (src/dynamics/solver/solver_body/plugin.rs:173-251) and `update_aabb`
/// (src/collision/collider/backend.rs:498-624) read.
fn gpu_upload_bodies(
    mut w: ResMut<Mi355xWorld>, mut st: ResMut<Mi355xStaging>, bodies: Query<crate::staging::BodyItem<'static>>,
    increments: Query<&VelocityIntegrationData>, local_accelerations: Query<&AccumulatedLocalAcceleration>,
    colliders: Query<(Entity, &Collider, &ColliderOf, &CollisionLayers, Option<&CollisionMargin>, Option<&SpeculativeMargin>, Has<Sensor>, Has<CollisionEventsEnabled>, Option<&ActiveCollisionHooks>, Option<&ColliderTransform>), Without<ColliderDisabled>>,   // (the filter of the reference's intervals and narrow phase: src/collision/broad_phase.rs:84,225,297, narrow_phase/system_param.rs:70)
    newly_disabled_colliders: Query<Entity, Added<ColliderDisabled>>,   // leave like despawned ones (remove_collider_on::<Add, (Disabled, ColliderDisabled)>, narrow_phase/mod.rs:153); they come back as new colliders when the component is removed (broad_phase.rs:117-130)
    mut removed_bodies: RemovedComponents<RigidBody>, mut removed_colliders: RemovedComponents<ColliderMarker>,
    js: Res<crate::joints::JointStaging>,
    live_joints: Query<(), (Or<(With<FixedJoint>, With<RevoluteJoint>, With<SphericalJoint>, With<PrismaticJoint>, With<DistanceJoint>)>, Without<JointDisabled>)>,
    mut host_shapes: ResMut<crate::host_shapes::HostShapeTable>, mut host_shapes_registered: Local<bool>,
) {
    if !*host_shapes_registered {   // (the resource's address is stable from here on: Bevy boxes resources)
        let raw = w.raw();
        let s = host_shapes.register(raw); w.check(s);
        *host_shapes_registered = true;
    }
    let st = &mut st.0;
    // Despawns since the last step, in the order Bevy reports the removals (= the order Avian's own observers ran in).  The staging still holds
    // LAST frame's numbering here: body index = position in `body_entities` (sorted by Entity, so dropping entries is the stable compaction
    // avn_despawn renumbers by), collider = Entity::index().  In the device closed loop the library removes them from its ContactGraph /
    // ConstraintGraph / islands without restarting the loop (round 4; it used to be avn_pipeline_enable(0 / 1): a step without warm starting).
    let gone_bodies: Vec<u32> = removed_bodies.read().filter_map(|e| st.body_index.get(&e).map(|&i| i as u32)).collect();
    let gone_of_bodies: std::collections::HashSet<u32> = gone_bodies.iter().copied().collect();
    let gone_colliders: Vec<u32> = removed_colliders.read().chain(newly_disabled_colliders.iter())
        .filter(|e| st.collider_slot.get(&e.index()).is_some_and(|&s| !gone_of_bodies.contains(&(st.c_body[s] as u32))))   // (a body's own colliders leave with it)
        .map(|e| e.index()).collect();
    // joints of the last upload that are gone: their entity lost its joint component (or gained JointDisabled), or one of their bodies was despawned
    let gone_joints: Vec<u32> = js.entities.iter().enumerate()
        .filter(|(i, e)| !live_joints.contains(**e) || gone_of_bodies.contains(&(js.body1[*i] as u32)) || gone_of_bodies.contains(&(js.body2[*i] as u32)))
        .map(|(i, _)| i as u32).collect();
    w.despawn(&gone_bodies, &gone_colliders, &gone_joints);
    st.fill_bodies(bodies.iter(), |e| increments.get(e).map_or((Vec3::ZERO, Vec3::ZERO), |v| (v.linear_increment(), v.angular_increment())));
    st.fill_colliders(colliders.iter(), &mut host_shapes);   // Ball / Cuboid: device shapes; everything else: AVN_SHAPE_HOST + its Collider into the table
    let (b, c) = (st.bodies_desc(), st.colliders_desc());
    let raw = w.raw();
    let s1 = unsafe { ffi::avn_bodies_upload(raw, &b) }; w.check(s1);
    let s2 = unsafe { ffi::avn_colliders_upload(raw, &c) }; w.check(s2);
    // apply_local_acceleration (ForceSystems::ApplyLocalAcceleration, a SubstepSchedule system in front of integrate_velocities): the device runs it inside
    // integrate_velocities from what ForcePlugin accumulated for this step; bodies_upload keeps the previous values only for an unchanged body count
    st.fill_local_accelerations(|e| local_accelerations.get(e).ok().map(|a| (a.linear, a.angular)));
    let s4 = if st.local_accel_any { unsafe { ffi::avn_local_accelerations_upload(raw, st.body_entities.len() as u32, st.local_accel_linear.as_ptr().cast(), st.local_accel_angular.as_ptr().cast()) } }
             else { unsafe { ffi::avn_local_accelerations_upload(raw, 0, core::ptr::null(), core::ptr::null()) } };
    w.check(s4);
    // compound bodies: the ColliderTransform of the colliders that are child entities (the device computes their Position / Rotation from the bodies': update_child_collider_position)
    if st.c_is_child.iter().any(|&c| c != 0) { let t = st.collider_transforms_desc(); let s3 = unsafe { ffi::avn_collider_transforms_upload(raw, &t) }; w.check(s3); }
}

/// BroadPhasePlugin replacement (src/collision/broad_phase.rs:347-474): device AABB update + sweep-and-prune; the new pairs come back in
/// the reference's emission order and become `ContactEdge`s exactly as `sweep_and_prune` creates them (:443-468).
fn gpu_broad_phase(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, mut settings: ResMut<Mi355xSettings>, mut contact_graph: ResMut<ContactGraph>,
    mut diagnostics: ResMut<CollisionDiagnostics>,
) {
    let mode = settings.effective_mode(&st.0);
    w.set_closed_loop(mode == Mi355xMode::ClosedLoop);      // avn_pipeline_enable(1 / 0) when the effective mode changed
    if mode == Mi355xMode::ClosedLoop { return; }   // avn_step runs the broad phase itself (gpu_solver)
    w.run_system(ffi::AVN_SYS_UPDATE_AABB);
    w.run_system(ffi::AVN_SYS_COLLECT_COLLISION_PAIRS);
    // (an index map built with the upload: a linear `position()` per returned pair was O(n^2) at cfg2's 1.2 M first-frame pairs)
    let entity_of = |index: u32| st.0.collider_entities[*st.0.collider_slot.get(&index).expect("collider of a device pair")];
    let pairs: Vec<ffi::avn_pair> = w.pairs().to_vec();
    for pair in pairs {
        // AVN_PAIR_NEEDS_CUSTOM_FILTER: CollisionHooks::filter_pairs cannot be called from the device (src/collision/broad_phase.rs:431-439):
        // such pairs are created unfiltered and the narrow phase's hook pass removes them, like pairs whose filter changed its mind.
        let (c1, c2) = (entity_of(pair.collider1), entity_of(pair.collider2));
        let (b1, b2) = (st.0.body_entities[pair.body1 as usize], st.0.body_entities[pair.body2 as usize]);
        let mut edge = ContactEdge::new(c1, c2);
        edge.body1 = Some(b1); edge.body2 = Some(b2);
        contact_graph.add_edge_with(edge, |contact_pair| {
            contact_pair.body1 = Some(b1); contact_pair.body2 = Some(b2);
            contact_pair.flags.set(ContactPairFlags::GENERATE_CONSTRAINTS, pair.flags & ffi::AVN_PAIR_GENERATE_CONSTRAINTS != 0);
            contact_pair.flags.set(ContactPairFlags::MODIFY_CONTACTS, pair.flags & ffi::AVN_PAIR_MODIFY_CONTACTS != 0);
            contact_pair.flags.set(ContactPairFlags::CONTACT_EVENTS, pair.flags & ffi::AVN_PAIR_CONTACT_EVENTS != 0);
        });
    }
    diagnostics.broad_phase += Duration::from_secs_f64(w.diagnostics().broad_phase_ms * 1e-3);
}

/// `prepare_contact_constraints` replacement (src/dynamics/solver/plugin.rs:363-448), host narrow phase mode: the colour-major manifold set.
fn gpu_upload_constraints(mut w: ResMut<Mi355xWorld>, mut st: ResMut<Mi355xStaging>, mut settings: ResMut<Mi355xSettings>, constraint_graph: Res<ConstraintGraph>, contact_graph: Res<ContactGraph>) {
    if settings.effective_mode(&st.0) != Mi355xMode::HostNarrowPhase { return; }
    st.0.fill_manifolds(&constraint_graph, &contact_graph);
    let m = st.0.manifolds_desc();
    let raw = w.raw();
    let s = unsafe { ffi::avn_manifolds_upload(raw, &m) };
    w.check(s);
}

/// SolverSystems::PrepareSolverBodies .. Restitution in one call.  `SolverSchedulePlugin`'s own runner (src/dynamics/solver/schedule.rs:194-213)
/// still loops over the (now nearly empty) `SubstepSchedule`; user systems added there see host `SolverBody` state only if the application
/// opts into per-substep round trips (caveat).
fn gpu_solver<H: CollisionHooks + 'static>(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, mut settings: ResMut<Mi355xSettings>, hooks: StaticSystemParam<H>, mut commands: Commands,
) where
    for<'w, 's> SystemParamItem<'w, 's, H>: CollisionHooks,
{
    match settings.effective_mode(&st.0) {
        Mi355xMode::HostNarrowPhase => w.run_system(ffi::AVN_SYS_SOLVER),   // (Avian's own broad / narrow phase called the hooks)
        Mi355xMode::ClosedLoop | Mi355xMode::Auto => {   // (effective_mode never returns Auto) avn_pipeline_enable(1) was called when the mode was selected: the whole PhysicsSchedule pass of the path
            if st.0.colliders_with_hooks == 0 { w.step(); return; }
            // CollisionHooks::filter_pairs / modify_contacts are called back from inside avn_step for the pairs of ActiveCollisionHooks colliders (hooks.rs)
            let raw = w.raw();
            let hooks = hooks.into_inner();
            let mut ctx = crate::hooks::HookContext::<H> { hooks: &hooks, commands: &mut commands, staging: &st.0 };
            let s = ctx.register(raw); w.check(s);
            w.step();
            let s = crate::hooks::HookContext::<H>::unregister(raw); w.check(s);
        }
    }
}

/// `writeback_solver_bodies` (src/dynamics/solver/solver_body/plugin.rs:255-284) + `store_contact_impulses` (src/dynamics/solver/plugin.rs:722-755).
fn gpu_download(
    mut w: ResMut<Mi355xWorld>, mut st: ResMut<Mi355xStaging>, mut settings: ResMut<Mi355xSettings>,
    mut bodies: Query<(&mut Position, &mut Rotation, &mut LinearVelocity, &mut AngularVelocity), With<SolverBody>>, mut contact_graph: ResMut<ContactGraph>,
) {
    let st = &mut st.0;
    let out = st.bodies_out_desc();
    let raw = w.raw();
    let s = unsafe { ffi::avn_bodies_download(raw, &out) }; w.check(s);
    for (i, &e) in st.body_entities.iter().enumerate() {
        let Ok((mut p, mut r, mut lv, mut av)) = bodies.get_mut(e) else { continue };   // only bodies that own a SolverBody are written back
        p.0 = Vec3::from_slice(&st.out_position[3 * i..]); r.0 = Quat::from_slice(&st.out_rotation[4 * i..]);
        lv.0 = Vec3::from_slice(&st.out_linear_velocity[3 * i..]); av.0 = Vec3::from_slice(&st.out_angular_velocity[3 * i..]);
    }
    if settings.effective_mode(st) == Mi355xMode::HostNarrowPhase {
        let imp = st.impulses_out_desc();
        let s = unsafe { ffi::avn_impulses_download(raw, &imp) }; w.check(s);
        for (m, &(contact_id, manifold_index)) in st.m_handles.iter().enumerate() {
            let Some(manifold) = contact_graph.get_manifold_mut(ContactManifoldHandle { contact_id, manifold_index }) else {
                unreachable!("Contact manifold {manifold_index:?} for contact ID {contact_id:?} not found in contact graph.")
            };
            for (k, contact) in manifold.points.iter_mut().enumerate().take(ffi::AVN_MAX_MANIFOLD_POINTS as usize) {
                contact.warm_start_normal_impulse = st.out_warm_n[4 * m + k];
                contact.warm_start_tangent_impulse = Vec2::new(st.out_warm_t[2 * (4 * m + k)], st.out_warm_t[2 * (4 * m + k) + 1]);
                contact.normal_impulse = st.out_normal_impulse[4 * m + k];
            }
        }
    }
}

/// The device's event timers into the reference's own resources (src/dynamics/solver/diagnostics.rs:13-37, src/collision/diagnostics.rs:13-19).
fn gpu_diagnostics(mut w: ResMut<Mi355xWorld>, mut solver: ResMut<SolverDiagnostics>, mut collision: ResMut<CollisionDiagnostics>) {
    let d = w.diagnostics();
    let ms = |x: f64| Duration::from_secs_f64(x * 1e-3);
    solver.prepare_constraints += ms(d.prepare_constraints_ms);
    solver.update_velocity_increments += ms(d.update_velocity_increments_ms);
    solver.integrate_velocities += ms(d.integrate_velocities_ms);
    solver.warm_start += ms(if d.per_system_valid != 0 { d.warm_start_ms } else { d.substeps_ms });   // graph replay: the loop's total, undivided
    solver.solve_constraints += ms(d.solve_constraints_ms);
    solver.integrate_positions += ms(d.integrate_positions_ms);
    solver.relax_velocities += ms(d.relax_velocities_ms);
    solver.apply_restitution += ms(d.apply_restitution_ms);
    solver.finalize += ms(d.finalize_ms);
    solver.store_impulses += ms(d.store_impulses_ms);
    solver.contact_constraint_count = d.contact_constraint_count;
    collision.narrow_phase += ms(d.narrow_phase_ms);
    collision.contact_count = d.contact_count;
}

/// `update_sleeping_states` + the decision of `sleep_islands` on the device (`avn_sleep_update`); `SleepIslands` / `WakeIslands` -- the
/// commands that insert `Sleeping`, move contact pairs to the ContactGraph's sleeping set and pop / push constraint handles -- are
/// Avian's own (src/dynamics/solver/islands/sleeping.rs:300-520) and are queued here exactly where `sleep_islands` queues them.
fn gpu_sleeping(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, time_to_sleep: Res<TimeToSleep>, length_unit: Res<PhysicsLengthUnit>, time: Res<Time>,
    mut timers: Query<(&mut SleepTimer, &BodyIslandNode)>, thresholds: Query<(&SleepThreshold, Has<SleepingDisabled>)>,
    mut islands: ResMut<PhysicsIslands>, mut commands: Commands,
) {
    if w.is_closed_loop() { return; }   // the closed loop's sleeping is closed_loop::gpu_closed_loop_sleeping
    let st = &st.0;
    // per-body `SleepThreshold` / `SleepingDisabled` in body order (avn_sleep_params.body_*), the world-level pair is only the fallback
    let (mut lin, mut ang, mut off) = (Vec::with_capacity(st.body_entities.len()), Vec::with_capacity(st.body_entities.len()), Vec::with_capacity(st.body_entities.len()));
    for &e in st.body_entities.iter() {
        let (t, d) = thresholds.get(e).map_or((SleepThreshold::default(), false), |(t, d)| (*t, d));
        lin.push(t.linear); ang.push(t.angular); off.push(d as u8);
    }
    let stats = w.sleep_update(time_to_sleep.0, (SleepThreshold::default().linear, SleepThreshold::default().angular), time.delta_secs(), length_unit.0 as f64,
                               Some((&lin, &ang, &off)));
    if stats.n_resting_islands == 0 && stats.n_waking_islands == 0 && stats.n_awake_bodies == 0 { return; }
    let (timer, _label, rests, wakes) = w.sleep_state(st.body_entities.len());
    let (mut to_sleep, mut to_wake) = (Vec::new(), Vec::new());
    islands.split_candidate_sleep_timer = 0.0;
    for (i, &e) in st.body_entities.iter().enumerate() {
        let Ok((mut t, node)) = timers.get_mut(e) else { continue };
        t.0 = timer[i];
        if rests[i] != 0 { to_sleep.push(node.island_id); }
        if wakes[i] != 0 { to_wake.push(node.island_id); }
        // the sleepiest body of an island that still has to be split picks the split candidate (sleeping.rs:231-239); the split itself is
        // Avian's `split_island` in SolverSystems::Finalize of the next step, over the host's own contact graph
        if timer[i] >= time_to_sleep.0 && timer[i] > islands.split_candidate_sleep_timer
            && islands.get(node.island_id).is_some_and(|isl| isl.constraints_removed() > 0)
        {
            islands.split_candidate = Some(node.island_id);
            islands.split_candidate_sleep_timer = timer[i];
        }
    }
    to_sleep.sort_unstable(); to_sleep.dedup(); to_wake.sort_unstable(); to_wake.dedup();
    // (the reference additionally refuses to sleep an island with pending splits, `constraints_removed > 0`: PhysicsIslands still holds
    //  that counter on the host, so the filter is applied here)
    commands.queue(move |world: &mut World| {
        let islands = world.resource::<PhysicsIslands>();
        let ok: Vec<IslandId> = to_sleep.into_iter().filter(|id| islands.get(*id).is_some_and(|i| i.constraints_removed() == 0 && !i.is_sleeping())).collect();
        SleepIslands(ok).apply(world);
        WakeIslands(to_wake).apply(world);
    });
}
