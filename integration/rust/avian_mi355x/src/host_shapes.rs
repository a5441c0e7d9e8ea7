//! Host shapes: every `Collider` that is not a parry Ball / Cuboid stays in the device closed loop -- its two shape-dependent methods are answered here.
//!
//! The library uploads such a collider as `AVN_SHAPE_HOST` and calls back, per step, for exactly the two things of the reference's hot path that depend on the shape
//! (include/avian_mi355x.h "host shapes"):
//!
//! * `update_aabb` (src/collision/collider/backend.rs:498-624) calls `AnyCollider::aabb_with_context` / `swept_aabb_with_context` -> [`aabb_trampoline`];
//! * `update_contacts` (src/collision/narrow_phase/system_param.rs:700-712) calls `AnyCollider::contact_manifolds_with_context` -> [`manifolds_trampoline`].
//!
//! Margins, the speculative filter, `prune_points`, `match_contacts` (the warm-start impulses never leave HBM), `normal_speed` and the status change stay on the device
//! for those pairs too.  The trampolines run on the thread that called `avn_step`, between two stream synchronisations, with `n` queries per call.
//!
//! This file covers `Collider` (the parry-backed `SimpleCollider`, `Context = ()`).  A custom `AnyCollider` with a context needs the same two functions with its own
//! `SystemParam` fetched before `avn_step` -- the shape of the code does not change.

use avian3d::prelude::*;
use avian_mi355x_sys as ffi;
use bevy::prelude::*;
use bevy::utils::HashMap;
use std::os::raw::c_void;

/// `Entity::index()` of every host-shaped collider -> its shape.  Refilled by `Staging::fill_colliders` (a clone of the `Collider`: an `Arc` bump), read by the
/// trampolines while `avn_step` runs.
#[derive(Resource, Default)]
pub struct HostShapeTable {
    pub colliders: HashMap<u32, Collider>,
}

impl HostShapeTable {
    /// Registers the trampolines with the world; `self` must outlive the registration (it is a `Resource`: it lives as long as the `App`).
    pub fn register(&mut self, raw: *mut ffi::avn_world) -> ffi::avn_status {
        unsafe { ffi::avn_host_shapes_set(raw, Some(aabb_trampoline), Some(manifolds_trampoline), self as *mut HostShapeTable as *mut c_void) }
    }
}

fn v3(a: &[f32; 3]) -> Vec3 { Vec3::new(a[0], a[1], a[2]) }
fn q4(a: &[f32; 4]) -> Quat { Quat::from_xyzw(a[0], a[1], a[2], a[3]) }

/// `avn_host_aabb_fn`: one `ColliderAabb` per query -- `aabb` for the start pose, `swept_aabb` when the device predicted an end pose (positive speculative margin).
/// The device grows the box by `contact_tolerance + collision margin` itself (backend.rs:560,618).
pub unsafe extern "C" fn aabb_trampoline(user: *mut c_void, scalar_bits: u32, n: u32, queries: *const c_void, aabbs_out: *mut c_void) {
    assert_eq!(scalar_bits, 32, "avian3d is built with the f32 feature in this crate");
    let table = &*(user as *const HostShapeTable);
    let q = std::slice::from_raw_parts(queries as *const ffi::avn_host_aabb_query_f32, n as usize);
    let out = std::slice::from_raw_parts_mut(aabbs_out as *mut ffi::avn_host_aabb_f32, n as usize);
    for (q, out) in q.iter().zip(out.iter_mut()) {
        let collider = &table.colliders[&q.collider];
        let aabb: ColliderAabb = if q.swept != 0 {
            collider.swept_aabb(v3(&q.start_position), Rotation(q4(&q.start_rotation)), v3(&q.end_position), Rotation(q4(&q.end_rotation)))
        } else {
            collider.aabb(v3(&q.start_position), Rotation(q4(&q.start_rotation)))
        };
        *out = ffi::avn_host_aabb_f32 { min: aabb.min.to_array(), max: aabb.max.to_array() };
    }
}

/// `avn_host_manifolds_fn`: `Collider::contact_manifolds` (src/collision/collider/parry/contact_query.rs:156-261) per query; `manifolds[0]` goes back as the reference
/// returns it -- normal, `anchor1`, penetration, feature ids.  (`anchor2 = anchor1 + (position1 - position2)` is recomputed on the device.)
pub unsafe extern "C" fn manifolds_trampoline(user: *mut c_void, scalar_bits: u32, n: u32, queries: *const c_void, manifolds_out: *mut c_void) {
    assert_eq!(scalar_bits, 32, "avian3d is built with the f32 feature in this crate");
    let table = &*(user as *const HostShapeTable);
    let q = std::slice::from_raw_parts(queries as *const ffi::avn_host_manifold_query_f32, n as usize);
    let out = std::slice::from_raw_parts_mut(manifolds_out as *mut ffi::avn_host_manifold_f32, n as usize);
    let mut manifolds: Vec<ContactManifold> = Vec::new();
    for (q, out) in q.iter().zip(out.iter_mut()) {
        let (c1, c2) = (&table.colliders[&q.collider1], &table.colliders[&q.collider2]);
        manifolds.clear();
        c1.contact_manifolds(c2, v3(&q.position1), Rotation(q4(&q.rotation1)), v3(&q.position2), Rotation(q4(&q.rotation2)), q.max_contact_distance, &mut manifolds);
        out.point_count = 0;
        let Some(m) = manifolds.first() else { continue };   // convex shapes: one manifold per pair (the device rows hold manifold index 0)
        out.normal = m.normal.to_array();
        for (k, p) in m.points.iter().take(ffi::AVN_MAX_QUERY_POINTS as usize).enumerate() {
            out.anchor1[3 * k..3 * k + 3].copy_from_slice(&p.anchor1.to_array());
            out.penetration[k] = p.penetration;
            out.feature_id1[k] = p.feature_id1.0;
            out.feature_id2[k] = p.feature_id2.0;
            out.point_count = k as u32 + 1;
        }
    }
}
