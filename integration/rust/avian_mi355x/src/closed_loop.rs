//! What the device closed loop owes the ECS every step: collision events, `CollidingEntities`, and -- with sleeping on -- the `Sleeping`
//! components and `SleepTimer`s of the bodies its own island manager put to sleep or woke.
//!
//! In `Mi355xMode::ClosedLoop` the ContactGraph lives in HBM (`avn_pipeline_enable(1)`) and Avian's narrow phase walks an empty pair list: the
//! status loop of `NarrowPhase::update` (src/collision/narrow_phase/system_param.rs:141-389) -- the place where the reference writes
//! `CollisionStart` / `CollisionEnd` and maintains `CollidingEntities` -- runs on the device.  The library reports what that loop saw:
//! `avn_pairs_get` + `avn_pipeline_new_pair_ids_get` (the step's new ContactEdges with their ContactIds, emission order) and
//! `avn_contact_changes_get` (the status changes, ascending ContactId: the order the reference processes them in).  [`ContactMirror`] keeps
//! (collider1, collider2, body1, body2, events enabled, touching) per live ContactId from those two streams and replays the loop's event side:
//!
//! * `DISJOINT_AABB`: `CollisionEnd` when the edge was TOUCHING with CONTACT_EVENTS (:155-167), `remove_colliding_entities`, the edge goes;
//! * `STARTED_TOUCHING`: `CollisionStart` when events are enabled (:210-218), `add_colliding_entities`;
//! * `STOPPED_TOUCHING`: `CollisionEnd` (:265-273), `remove_colliding_entities`.
//!
//! Events are written in ascending ContactId per step, as the reference writes them.

use crate::{plugins::Mi355xStaging, world::Mi355xWorld};
use avian3d::prelude::*;
use avian_mi355x_sys as ffi;
use bevy::{platform::collections::HashMap, prelude::*};

#[derive(Clone, Copy)]
struct MirrorEdge { collider1: Entity, collider2: Entity, body1: Option<Entity>, body2: Option<Entity>, events: bool, touching: bool }

/// Host mirror of the device ContactGraph's edges, keyed by ContactId (only what events need).
#[derive(Resource, Default)]
pub struct ContactMirror { edges: HashMap<u32, MirrorEdge> }

impl ContactMirror {
    /// Leaving the closed loop (`avn_pipeline_enable(0)`) drops the device rows; the pairs that were touching end, as they would when their
    /// ContactEdges are removed (:155-167).
    pub fn clear(&mut self) { self.edges.clear(); }
    pub fn len(&self) -> usize { self.edges.len() }
    pub fn is_empty(&self) -> bool { self.edges.is_empty() }
}

fn add_colliding(q: &mut Query<&mut CollidingEntities>, a: Entity, b: Entity) {   // NarrowPhase::add_colliding_entities (:402-414)
    if let Ok(mut c) = q.get_mut(a) { c.insert(b); }
    if let Ok(mut c) = q.get_mut(b) { c.insert(a); }
}
fn remove_colliding(q: &mut Query<&mut CollidingEntities>, a: Entity, b: Entity) {   // ::remove_colliding_entities (:416-428)
    if let Ok(mut c) = q.get_mut(a) { c.remove(&b); }
    if let Ok(mut c) = q.get_mut(b) { c.remove(&a); }
}

/// Runs right after `gpu_solver` (the `avn_step` of the closed loop), inside `PhysicsStepSystems::NarrowPhase`' successor position the plugin
/// gives it (plugins.rs): `CollisionStart` / `CollisionEnd` messages and `CollidingEntities` of the step.
pub fn gpu_closed_loop_events(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, mut mirror: ResMut<ContactMirror>, mut colliding: Query<&mut CollidingEntities>,
    mut started: MessageWriter<CollisionStart>, mut ended: MessageWriter<CollisionEnd>,
) {
    if !w.is_closed_loop() { if !mirror.is_empty() { mirror.clear(); } return; }
    let st = &st.0;
    // (1) the step's new ContactEdges: add_edge_and_key_with in emission order (src/collision/broad_phase.rs:443-468)
    let ids: Vec<u32> = w.new_pair_ids().to_vec();
    let pairs: Vec<ffi::avn_pair> = w.pairs().to_vec();
    debug_assert_eq!(ids.len(), pairs.len());
    for (&id, pr) in ids.iter().zip(pairs.iter()) {
        let entity_of = |index: u32| st.collider_slot.get(&index).map(|&s| st.collider_entities[s]);
        let (Some(c1), Some(c2)) = (entity_of(pr.collider1), entity_of(pr.collider2)) else { continue };
        let body = |b: i32| if b >= 0 { st.body_entities.get(b as usize).copied() } else { None };
        mirror.edges.insert(id, MirrorEdge { collider1: c1, collider2: c2, body1: body(pr.body1), body2: body(pr.body2),
                                             events: pr.flags & ffi::AVN_PAIR_CONTACT_EVENTS != 0, touching: false });
    }
    // (2) the status changes, ascending ContactId
    let changes: Vec<ffi::avn_contact_change> = w.contact_changes().to_vec();
    for ch in changes {
        let Some(edge) = mirror.edges.get_mut(&ch.contact_id) else { continue };
        let e = *edge;
        if ch.flags & ffi::AVN_CP_DISJOINT_AABB != 0 {
            if e.touching && e.events { ended.write(CollisionEnd { collider1: e.collider1, collider2: e.collider2, body1: e.body1, body2: e.body2 }); }
            remove_colliding(&mut colliding, e.collider1, e.collider2);
            mirror.edges.remove(&ch.contact_id);
        } else if ch.flags & ffi::AVN_CP_STARTED_TOUCHING != 0 {
            if e.events { started.write(CollisionStart { collider1: e.collider1, collider2: e.collider2, body1: e.body1, body2: e.body2 }); }
            add_colliding(&mut colliding, e.collider1, e.collider2);
            edge.touching = true;
        } else if ch.flags & ffi::AVN_CP_STOPPED_TOUCHING != 0 {
            if e.events { ended.write(CollisionEnd { collider1: e.collider1, collider2: e.collider2, body1: e.body1, body2: e.body2 }); }
            remove_colliding(&mut colliding, e.collider1, e.collider2);
            edge.touching = false;
        }
    }
}

/// Sleeping in the closed loop.  Avian's own `PhysicsIslands` receives no contacts in this mode (its narrow phase sees no pair), so it must not
/// decide anything: the library's island manager does (`avn_sleeping_enable`: persistent islands, deferred split, SleepIslands / WakeIslands in the
/// reference's order, src/dynamics/solver/islands/{mod,sleeping}.rs), and this system mirrors its verdict into the ECS -- the `Sleeping` marker
/// (what `RigidBodyActiveFilter` and user queries read) and the `SleepTimer` -- after every step.  Per-body `SleepThreshold` / `SleepingDisabled`
/// travel with `avn_sleeping_enable`; bodies the application moved or kicked are woken through `avn_wake_bodies` (`wake_on_changed`,
/// sleeping.rs:556-604: `Changed<Position | Rotation | LinearVelocity | AngularVelocity | ...>` that did not come from the write-back).
#[derive(Resource, Default)]
pub struct ClosedLoopSleeping { enabled: bool, island: Vec<u32>, next: Vec<u32>, sleeping: Vec<u8>, timer: Vec<f32>, lin: Vec<f32>, ang: Vec<f32>, off: Vec<u8> }

#[allow(clippy::too_many_arguments)]
pub fn gpu_closed_loop_sleeping(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, mut state: ResMut<ClosedLoopSleeping>, time_to_sleep: Option<Res<TimeToSleep>>,
    length_unit: Res<PhysicsLengthUnit>, time: Res<Time>, thresholds: Query<(&SleepThreshold, Has<SleepingDisabled>)>,
    mut bodies: Query<(&mut SleepTimer, Has<Sleeping>)>, mut commands: Commands,
) {
    let st = &st.0;
    let want = w.is_closed_loop() && time_to_sleep.is_some();   // (no TimeToSleep resource: the application runs without Avian's sleeping plugin)
    if want != state.enabled {
        if want {
            let n = st.body_entities.len();
            state.lin.clear(); state.ang.clear(); state.off.clear();
            for &e in st.body_entities.iter() {
                let (t, d) = thresholds.get(e).map_or((SleepThreshold::default(), false), |(t, d)| (*t, d));
                state.lin.push(t.linear); state.ang.push(t.angular); state.off.push(d as u8);
            }
            debug_assert_eq!(state.lin.len(), n);
            let d = SleepThreshold::default();
            w.set_sleeping(Some(ffi::avn_sleep_params {
                struct_size: core::mem::size_of::<ffi::avn_sleep_params>() as u32, time_to_sleep: time_to_sleep.as_ref().unwrap().0, linear_threshold: d.linear,
                angular_threshold: d.angular, delta_secs: time.delta_secs(), length_unit: length_unit.0 as f64,
                body_linear_threshold: state.lin.as_ptr(), body_angular_threshold: state.ang.as_ptr(), body_sleeping_disabled: state.off.as_ptr(),
            }));
        } else if w.is_closed_loop() { w.set_sleeping(None); }
        state.enabled = want;
    }
    if !state.enabled { return; }
    let n = st.body_entities.len();
    state.island.resize(n, 0); state.next.resize(n, 0); state.sleeping.resize(n, 0); state.timer.resize(n, 0.0);
    let out = ffi::avn_sleeping_out { island: state.island.as_mut_ptr(), next_in_island: state.next.as_mut_ptr(), sleeping: state.sleeping.as_mut_ptr(),
                                      sleep_timer: state.timer.as_mut_ptr() };
    let raw = w.raw();
    let s = unsafe { ffi::avn_sleeping_state_get(raw, &out) };
    w.check(s);
    for (i, &e) in st.body_entities.iter().enumerate() {
        let Ok((mut timer, is_sleeping)) = bodies.get_mut(e) else { continue };
        timer.0 = state.timer[i];
        match (state.sleeping[i] != 0, is_sleeping) {
            (true, false) => { commands.entity(e).insert(Sleeping); }     // SleepIslands (sleeping.rs:300-420): the body's marker
            (false, true) => { commands.entity(e).remove::<Sleeping>(); } // WakeIslands (:438-520)
            _ => {}
        }
    }
}

/// `wake_on_changed` (sleeping.rs:556-604) for the closed loop.  A sleeping body owns no `SolverBody` (Avian's SolverBodyPlugin removes it with the
/// `Sleeping` marker), so `gpu_download` never writes it: any change to its transform, velocities or constant forces since the last step is the
/// application's, and wakes it (`avn_wake_bodies` = the `WakeBody` command on the library's island manager).
#[allow(clippy::type_complexity)]
pub fn gpu_closed_loop_wake_on_changed(
    mut w: ResMut<Mi355xWorld>, st: Res<Mi355xStaging>, state: Res<ClosedLoopSleeping>,
    changed: Query<Entity, (With<Sleeping>, Or<(Changed<Position>, Changed<Rotation>, Changed<LinearVelocity>, Changed<AngularVelocity>, Changed<ConstantForce>, Changed<ConstantTorque>)>)>,
) {
    if !state.enabled { return; }
    let wake: Vec<u32> = changed.iter().filter_map(|e| st.0.body_index.get(&e).map(|&i| i as u32)).collect();
    if !wake.is_empty() { w.wake_bodies(&wake); }
}
