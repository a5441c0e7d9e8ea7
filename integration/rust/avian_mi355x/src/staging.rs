//! ECS -> Structure-of-Arrays staging.  The C ABI borrows host arrays for the duration of a call (include/avian_mi355x.h, "Ownership");
//! these buffers are reused across steps so that a step allocates nothing.  Body index = position in `body_entities`, which is the
//! order of the body query sorted by `Entity` (stable across steps; the solver's results do not depend on it, the broad phase's pair
//! ORDER follows the collider upload order exactly like `AabbIntervals` follows insertion order, src/collision/broad_phase.rs:296-315).

use avian3d::prelude::*;
use avian_mi355x_sys as ffi;
use bevy::{ecs::entity::EntityHashMap, prelude::*};
use core::ffi::c_void;

#[derive(Default)]
pub struct Staging {
    pub body_entities: Vec<Entity>,
    pub body_index: EntityHashMap<i32>,
    // avn_bodies
    pub position: Vec<f32>, pub rotation: Vec<f32>, pub linear_velocity: Vec<f32>, pub angular_velocity: Vec<f32>,
    pub inv_mass: Vec<f32>, pub inv_inertia_local: Vec<f32>, pub center_of_mass: Vec<f32>,
    pub linear_damping: Vec<f32>, pub angular_damping: Vec<f32>, pub gravity_scale: Vec<f32>,
    pub accel_linear: Vec<f32>, pub accel_angular: Vec<f32>, pub max_linear_speed: Vec<f32>, pub max_angular_speed: Vec<f32>,
    pub rb_type: Vec<u8>, pub locked_axes: Vec<u8>, pub dominance: Vec<i8>, pub body_flags: Vec<u8>,
    /// `AccumulatedLocalAcceleration` per body (src/dynamics/rigid_body/forces/mod.rs:661-673), for `avn_local_accelerations_upload`; `local_accel_any`:
    /// some body's is non-zero this step (all zero = the call is made with count 0 and the substep system costs nothing)
    pub local_accel_linear: Vec<f32>, pub local_accel_angular: Vec<f32>, pub local_accel_any: bool,
    // avn_colliders
    pub collider_entities: Vec<Entity>,
    /// `Entity::index()` of a collider -> its slot of the last upload (what a device pair's collider index resolves through)
    pub collider_slot: bevy::platform::collections::HashMap<u32, usize>,
    /// colliders of the last upload that carry `ActiveCollisionHooks` (filter_pairs / modify_contacts need the host narrow phase)
    pub colliders_with_hooks: usize,
    /// colliders the closed loop cannot take (not on a body the staging knows, or a COMPOSITE shape -- several manifolds per pair): with any of them the
    /// step must keep Avian's own narrow phase (`Mi355xSettings::effective_mode`)
    pub colliders_unsupported: usize,
    pub c_entity_index: Vec<u32>, pub c_body: Vec<i32>, pub c_shape: Vec<u8>, pub c_half_extents: Vec<f32>,
    pub c_memberships: Vec<u32>, pub c_filters: Vec<u32>, pub c_flags: Vec<u8>, pub c_margin: Vec<f32>, pub c_speculative: Vec<f32>,
    /// child colliders (a collider entity that is not its rigid body's entity): `ColliderTransform::translation` / `rotation`; the device computes the child's
    /// Position / Rotation from its body's every step, as `update_child_collider_position` does (src/collision/collider/collider_transform/plugin.rs:62-91)
    pub c_is_child: Vec<u8>, pub c_local_translation: Vec<f32>, pub c_local_rotation: Vec<f32>,
    // avn_manifolds (colour-major: the order of GraphColor::manifold_handles, src/dynamics/solver/constraint_graph.rs:66-80)
    pub m_offsets: [u32; ffi::AVN_GRAPH_COLOR_COUNT as usize + 1],
    pub m_handles: Vec<(ContactId, usize)>,
    pub m_body1: Vec<i32>, pub m_body2: Vec<i32>, pub m_normal: Vec<f32>, pub m_friction: Vec<f32>, pub m_restitution: Vec<f32>,
    pub m_tangent_velocity: Vec<f32>, pub m_point_count: Vec<u8>, pub m_flags: Vec<u8>,
    pub m_anchor1: Vec<f32>, pub m_anchor2: Vec<f32>, pub m_penetration: Vec<f32>, pub m_normal_speed: Vec<f32>,
    pub m_warm_n: Vec<f32>, pub m_warm_t: Vec<f32>,
    // downloads
    pub out_position: Vec<f32>, pub out_rotation: Vec<f32>, pub out_linear_velocity: Vec<f32>, pub out_angular_velocity: Vec<f32>,
    pub out_warm_n: Vec<f32>, pub out_warm_t: Vec<f32>, pub out_normal_impulse: Vec<f32>,
}

fn p<T>(v: &[T]) -> *const c_void { if v.is_empty() { core::ptr::null() } else { v.as_ptr() as *const c_void } }
fn pm<T>(v: &mut [T]) -> *mut c_void { if v.is_empty() { core::ptr::null_mut() } else { v.as_mut_ptr() as *mut c_void } }
fn push3(v: &mut Vec<f32>, a: Vec3) { v.extend_from_slice(&[a.x, a.y, a.z]); }

/// What `prepare_solver_bodies` / `pre_process_velocity_increments` / `update_aabb` read per rigid body
/// (src/dynamics/solver/solver_body/plugin.rs:173-251, src/dynamics/integrator/mod.rs:260-313).
pub type BodyItem<'a> = (
    Entity, &'a RigidBody, &'a Position, &'a Rotation, &'a LinearVelocity, &'a AngularVelocity, &'a ComputedMass, &'a ComputedAngularInertia,
    &'a ComputedCenterOfMass, Option<&'a LinearDamping>, Option<&'a AngularDamping>, Option<&'a GravityScale>, Option<&'a LockedAxes>,
    Option<&'a Dominance>, Option<&'a MaxLinearSpeed>, Option<&'a MaxAngularSpeed>, Has<Sleeping>, Has<RigidBodyDisabled>,
);

impl Staging {
    pub fn fill_bodies<'a>(&mut self, bodies: impl Iterator<Item = BodyItem<'a>>, accelerations: impl Fn(Entity) -> (Vec3, Vec3)) {
        macro_rules! clear { ($($f:ident),*) => { $( self.$f.clear(); )* } }
        clear!(body_entities, position, rotation, linear_velocity, angular_velocity, inv_mass, inv_inertia_local, center_of_mass, linear_damping,
               angular_damping, gravity_scale, accel_linear, accel_angular, max_linear_speed, max_angular_speed, rb_type, locked_axes, dominance, body_flags);
        self.body_index.clear();
        let mut items: Vec<BodyItem<'a>> = bodies.collect();
        items.sort_by_key(|it| it.0);
        for (e, rb, pos, rot, lv, av, mass, inertia, com, ld, ad, gs, locked, dom, mls, mas, sleeping, disabled) in items {
            self.body_index.insert(e, self.body_entities.len() as i32);
            self.body_entities.push(e);
            push3(&mut self.position, pos.0); self.rotation.extend_from_slice(&rot.0.to_array());
            push3(&mut self.linear_velocity, lv.0); push3(&mut self.angular_velocity, av.0);
            self.inv_mass.push(mass.inverse());
            let i = inertia.inverse();   // local inverse tensor, symmetric: m00 m01 m02 m11 m12 m22
            self.inv_inertia_local.extend_from_slice(&[i.m00, i.m01, i.m02, i.m11, i.m12, i.m22]);
            push3(&mut self.center_of_mass, com.0);
            self.linear_damping.push(ld.map_or(0.0, |d| d.0)); self.angular_damping.push(ad.map_or(0.0, |d| d.0));
            self.gravity_scale.push(gs.map_or(1.0, |g| g.0));
            let (al, aa) = accelerations(e);   // AccumulatedLocalAcceleration / constant forces already converted by ForcePlugin
            push3(&mut self.accel_linear, al); push3(&mut self.accel_angular, aa);
            self.max_linear_speed.push(mls.map_or(f32::INFINITY, |m| m.0)); self.max_angular_speed.push(mas.map_or(f32::INFINITY, |m| m.0));
            self.rb_type.push(match rb { RigidBody::Dynamic => ffi::AVN_RB_DYNAMIC, RigidBody::Static => ffi::AVN_RB_STATIC, RigidBody::Kinematic => ffi::AVN_RB_KINEMATIC } as u8);
            self.locked_axes.push(locked.map_or(0, |l| l.to_bits()));
            self.dominance.push(dom.map_or(0, |d| d.0));
            self.body_flags.push((if sleeping { ffi::AVN_BODY_SLEEPING } else { 0 } | if disabled { ffi::AVN_BODY_DISABLED } else { 0 }) as u8);
        }
    }

    /// What `apply_local_acceleration` reads (src/dynamics/rigid_body/forces/plugin.rs:207-241), in the body order of the last `fill_bodies`:
    /// ForcePlugin's own systems have accumulated ConstantLocalForce & co. and `Forces::apply_local_*` into the component before the step (:145-203) and
    /// clear it after (:243-251) -- both stay Avian's; only the substep system that consumes it runs on the device.
    pub fn fill_local_accelerations(&mut self, get: impl Fn(Entity) -> Option<(Vec3, Vec3)>) {
        self.local_accel_linear.clear(); self.local_accel_angular.clear(); self.local_accel_any = false;
        for &e in &self.body_entities {
            let (l, a) = get(e).unwrap_or((Vec3::ZERO, Vec3::ZERO));
            self.local_accel_any |= l != Vec3::ZERO || a != Vec3::ZERO;
            push3(&mut self.local_accel_linear, l); push3(&mut self.local_accel_angular, a);
        }
    }

    pub fn bodies_desc(&self) -> ffi::avn_bodies {
        ffi::avn_bodies {
            count: self.body_entities.len() as u32,
            position: p(&self.position), rotation: p(&self.rotation), linear_velocity: p(&self.linear_velocity), angular_velocity: p(&self.angular_velocity),
            inv_mass: p(&self.inv_mass), inv_inertia_local: p(&self.inv_inertia_local), center_of_mass: p(&self.center_of_mass),
            linear_damping: p(&self.linear_damping), angular_damping: p(&self.angular_damping), gravity_scale: p(&self.gravity_scale),
            accel_linear: p(&self.accel_linear), accel_angular: p(&self.accel_angular),
            max_linear_speed: p(&self.max_linear_speed), max_angular_speed: p(&self.max_angular_speed),
            rb_type: self.rb_type.as_ptr(), locked_axes: self.locked_axes.as_ptr(), dominance: self.dominance.as_ptr(), body_flags: self.body_flags.as_ptr(),
        }
    }

    /// Colliders in the order `AabbIntervals` would hold them (existing ones keep their place, new ones are appended:
    /// src/collision/broad_phase.rs:214-315).  Ball / Cuboid shapes have device kernels; every other shape is uploaded as `AVN_SHAPE_HOST` and answered by the two
    /// callbacks of host_shapes.rs (its `Collider` goes into `host_shapes`): it stays in the closed loop.  Only a collider that is not on a known rigid body is unsupported.
    pub fn fill_colliders<'a>(
        &mut self,
        colliders: impl Iterator<Item = (Entity, &'a Collider, &'a ColliderOf, &'a CollisionLayers, Option<&'a CollisionMargin>, Option<&'a SpeculativeMargin>, Has<Sensor>, Has<CollisionEventsEnabled>, Option<&'a ActiveCollisionHooks>, Option<&'a ColliderTransform>)>,
        host_shapes: &mut crate::host_shapes::HostShapeTable,
    ) {
        host_shapes.colliders.clear();
        macro_rules! clear { ($($f:ident),*) => { $( self.$f.clear(); )* } }
        clear!(collider_entities, collider_slot, c_entity_index, c_body, c_shape, c_half_extents, c_memberships, c_filters, c_flags, c_margin, c_speculative, c_is_child, c_local_translation, c_local_rotation);
        self.colliders_with_hooks = 0;
        self.colliders_unsupported = 0;
        for (e, collider, of, layers, margin, spec, sensor, events, hooks, transform) in colliders {
            let shape = collider.shape_scaled();
            let (kind, he) = if let Some(b) = shape.as_ball() { (ffi::AVN_SHAPE_BALL, Vec3::new(b.radius, 0.0, 0.0)) }
                             else if let Some(c) = shape.as_cuboid() { (ffi::AVN_SHAPE_CUBOID, Vec3::new(c.half_extents.x, c.half_extents.y, c.half_extents.z)) }
                             else { (ffi::AVN_SHAPE_HOST, Vec3::ZERO) };   // capsule, cylinder, cone, convex hull, ...: aabb / contact_manifolds stay on the host
            let Some(&body) = self.body_index.get(&of.body) else { self.colliders_unsupported += 1; continue };
            // A composite shape (TriMesh, HeightField, Compound, Polyline, Voxels) answers contact_manifolds with SEVERAL manifolds per pair, each pushed into a colour of
            // its own (src/dynamics/solver/constraint_graph.rs:162-236); the device's contact rows hold one manifold per pair (include/avian_mi355x.h, host shapes: "convex
            // shapes"), so such a world keeps Avian's narrow phase and ContactGraph (HostNarrowPhase mode: every manifold of every pair is uploaded) instead of silently
            // solving manifolds[0] only.
            if kind == ffi::AVN_SHAPE_HOST && shape.as_composite_shape().is_some() { self.colliders_unsupported += 1; }
            if kind == ffi::AVN_SHAPE_HOST { host_shapes.colliders.insert(e.index(), collider.clone()); }
            self.collider_slot.insert(e.index(), self.collider_entities.len());
            // ActiveCollisionHooks (src/collision/hooks.rs:213-231): which of the two hooks this collider asks for (broad_phase.rs:266-273)
            let filter = hooks.is_some_and(|h| h.contains(ActiveCollisionHooks::FILTER_PAIRS));
            let modify = hooks.is_some_and(|h| h.contains(ActiveCollisionHooks::MODIFY_CONTACTS));
            if filter || modify { self.colliders_with_hooks += 1; }
            self.collider_entities.push(e);
            self.c_entity_index.push(e.index()); self.c_body.push(body); self.c_shape.push(kind as u8); push3(&mut self.c_half_extents, he);
            self.c_memberships.push(layers.memberships.0); self.c_filters.push(layers.filters.0);
            self.c_flags.push((if sensor { ffi::AVN_COLLIDER_SENSOR } else { 0 } | if events { ffi::AVN_COLLIDER_EVENTS } else { 0 }
                               | if filter { ffi::AVN_COLLIDER_FILTER_PAIRS } else { 0 } | if modify { ffi::AVN_COLLIDER_MODIFY_CONTACTS } else { 0 }) as u8);
            self.c_margin.push(margin.map_or(0.0, |m| m.0)); self.c_speculative.push(spec.map_or(-1.0, |s| s.0));   // < 0: absent (NarrowPhaseConfig default applies)
            // a child collider: ColliderOf names another entity than its own (update_child_collider_position's query is `Without<RigidBody>`); the shape above is
            // already scaled (`shape_scaled`), the translation carries the parents' scale (propagate_collider_transforms)
            let child = of.body != e;
            let t = transform.copied().unwrap_or_default();
            self.c_is_child.push(child as u8);
            push3(&mut self.c_local_translation, t.translation);
            self.c_local_rotation.extend_from_slice(&t.rotation.0.to_array());
        }
    }

    /// `avn_collider_transforms_upload`, after every `avn_colliders_upload` (which puts all colliders back on their bodies).
    pub fn collider_transforms_desc(&self) -> ffi::avn_collider_transforms {
        ffi::avn_collider_transforms {
            count: self.collider_entities.len() as u32, is_child: self.c_is_child.as_ptr(), translation: p(&self.c_local_translation), rotation: p(&self.c_local_rotation),
        }
    }

    pub fn colliders_desc(&self) -> ffi::avn_colliders {
        ffi::avn_colliders {
            count: self.collider_entities.len() as u32, entity_index: self.c_entity_index.as_ptr(), body: self.c_body.as_ptr(), shape: self.c_shape.as_ptr(),
            half_extents: p(&self.c_half_extents), memberships: self.c_memberships.as_ptr(), filters: self.c_filters.as_ptr(), collider_flags: self.c_flags.as_ptr(),
            collision_margin: p(&self.c_margin), speculative_margin: p(&self.c_speculative),
        }
    }

    /// The manifolds the solver works on, colour-major: exactly the walk of `prepare_contact_constraints`
    /// (src/dynamics/solver/plugin.rs:363-448) over `ConstraintGraph.colors[c].manifold_handles`, resolved through `ContactGraph::get_by_id`.
    pub fn fill_manifolds(&mut self, constraint_graph: &ConstraintGraph, contact_graph: &ContactGraph) {
        macro_rules! clear { ($($f:ident),*) => { $( self.$f.clear(); )* } }
        clear!(m_handles, m_body1, m_body2, m_normal, m_friction, m_restitution, m_tangent_velocity, m_point_count, m_flags, m_anchor1, m_anchor2,
               m_penetration, m_normal_speed, m_warm_n, m_warm_t);
        for (c, color) in constraint_graph.colors.iter().enumerate() {
            self.m_offsets[c] = self.m_handles.len() as u32;
            for handle in &color.manifold_handles {
                let pair = contact_graph.get_by_id(handle.contact_id).unwrap_or_else(|| panic!("Contact pair not found in graph: {:?}", handle.contact_id)).1;
                let manifold = &pair.manifolds[handle.manifold_index];
                let (Some(b1), Some(b2)) = (pair.body1, pair.body2) else { continue };
                let (Some(&i1), Some(&i2)) = (self.body_index.get(&b1), self.body_index.get(&b2)) else { continue };
                self.m_handles.push((handle.contact_id, handle.manifold_index));
                self.m_body1.push(i1); self.m_body2.push(i2);
                push3(&mut self.m_normal, manifold.normal); self.m_friction.push(manifold.friction); self.m_restitution.push(manifold.restitution);
                push3(&mut self.m_tangent_velocity, manifold.tangent_velocity);
                let n = manifold.points.len().min(ffi::AVN_MAX_MANIFOLD_POINTS as usize);
                self.m_point_count.push(n as u8);
                self.m_flags.push(if pair.generates_constraints() { ffi::AVN_MANIFOLD_GENERATES_CONSTRAINTS as u8 } else { 0 });
                for k in 0..ffi::AVN_MAX_MANIFOLD_POINTS as usize {   // slot = 4 * manifold + point, unused points zero
                    let pt = manifold.points.get(k).filter(|_| k < n);
                    push3(&mut self.m_anchor1, pt.map_or(Vec3::ZERO, |q| q.anchor1)); push3(&mut self.m_anchor2, pt.map_or(Vec3::ZERO, |q| q.anchor2));
                    self.m_penetration.push(pt.map_or(0.0, |q| q.penetration)); self.m_normal_speed.push(pt.map_or(0.0, |q| q.normal_speed));
                    self.m_warm_n.push(pt.map_or(0.0, |q| q.warm_start_normal_impulse));
                    let t = pt.map_or(Vec2::ZERO, |q| q.warm_start_tangent_impulse);
                    self.m_warm_t.extend_from_slice(&[t.x, t.y]);
                }
            }
        }
        self.m_offsets[ffi::AVN_GRAPH_COLOR_COUNT as usize] = self.m_handles.len() as u32;
    }

    pub fn manifolds_desc(&self) -> ffi::avn_manifolds {
        ffi::avn_manifolds {
            count: self.m_handles.len() as u32, color_offsets: self.m_offsets.as_ptr(), body1: self.m_body1.as_ptr(), body2: self.m_body2.as_ptr(),
            normal: p(&self.m_normal), friction: p(&self.m_friction), restitution: p(&self.m_restitution), tangent_velocity: p(&self.m_tangent_velocity),
            point_count: self.m_point_count.as_ptr(), manifold_flags: self.m_flags.as_ptr(), anchor1: p(&self.m_anchor1), anchor2: p(&self.m_anchor2),
            penetration: p(&self.m_penetration), normal_speed: p(&self.m_normal_speed), warm_start_normal_impulse: p(&self.m_warm_n),
            warm_start_tangent_impulse: p(&self.m_warm_t),
        }
    }

    pub fn bodies_out_desc(&mut self) -> ffi::avn_bodies_out {
        let n = self.body_entities.len();
        self.out_position.resize(3 * n, 0.0); self.out_rotation.resize(4 * n, 0.0);
        self.out_linear_velocity.resize(3 * n, 0.0); self.out_angular_velocity.resize(3 * n, 0.0);
        ffi::avn_bodies_out { position: pm(&mut self.out_position), rotation: pm(&mut self.out_rotation), linear_velocity: pm(&mut self.out_linear_velocity),
                              angular_velocity: pm(&mut self.out_angular_velocity) }
    }

    pub fn impulses_out_desc(&mut self) -> ffi::avn_impulses_out {
        let m = self.m_handles.len();
        self.out_warm_n.resize(4 * m, 0.0); self.out_warm_t.resize(8 * m, 0.0); self.out_normal_impulse.resize(4 * m, 0.0);
        ffi::avn_impulses_out { warm_start_normal_impulse: pm(&mut self.out_warm_n), warm_start_tangent_impulse: pm(&mut self.out_warm_t),
                                normal_impulse: pm(&mut self.out_normal_impulse) }
    }
}
