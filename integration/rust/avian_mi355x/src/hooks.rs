//! `CollisionHooks` inside the device closed loop: `filter_pairs` and `modify_contacts` answered here, through `avn_collision_hooks_set`.
//!
//! Avian calls the two hooks (src/collision/hooks.rs:137-231) only for colliders that carry `ActiveCollisionHooks`:
//!
//! * `sweep_and_prune` (src/collision/broad_phase.rs:431-439) asks `filter_pairs` about every NEW candidate pair one of whose colliders has `FILTER_PAIRS`, before
//!   the pair becomes a `ContactEdge` -> [`filter_trampoline`], `n` pairs per call in the sweep's emission order;
//! * `update_contacts` (src/collision/narrow_phase/system_param.rs:770-778) shows `modify_contacts` every touching `ContactPair` flagged `MODIFY_CONTACTS`, after the
//!   speculative filter and `prune_points`, before `match_contacts` and the status change -> [`modify_trampoline`], `n` records per call in ascending `ContactId`.
//!
//! Everything on either side of the hooks stays on the device (include/avian_mi355x.h "collision hooks"); per step only the asked-about pairs (12 B each) and the
//! shown `ContactPair`s (232 B each way) cross the bus.  Until round 6 a single `ActiveCollisionHooks` collider sent the whole world to `HostNarrowPhase` mode.
//!
//! The trampolines run on the thread that called `avn_step` -- inside `plugins::gpu_solver::<H>`, which owns the `StaticSystemParam<H>` and the `Commands` the hooks
//! take -- between two stream synchronisations of the library.

use avian3d::prelude::*;
use avian_mi355x_sys as ffi;
use bevy::ecs::system::SystemParamItem;
use bevy::prelude::*;
use std::os::raw::c_void;

use crate::staging::Staging;

/// What the trampolines need for the duration of ONE `avn_step`: the user's hooks, the step's `Commands`, and the index -> `Entity` tables of the last upload.
/// Built on `gpu_solver`'s stack, registered before the step and unregistered after it (the pointers do not outlive the system run).
pub struct HookContext<'a, 'w, 's, H: CollisionHooks + 'static>
where
    for<'x, 'y> SystemParamItem<'x, 'y, H>: CollisionHooks,
{
    pub hooks: &'a SystemParamItem<'w, 's, H>,
    pub commands: &'a mut Commands<'w, 's>,
    pub staging: &'a Staging,
}

impl<'a, 'w, 's, H: CollisionHooks + 'static> HookContext<'a, 'w, 's, H>
where
    for<'x, 'y> SystemParamItem<'x, 'y, H>: CollisionHooks,
{
    /// `avn_collision_hooks_set` with this context as `user`.  Call [`HookContext::unregister`] before the context goes out of scope.
    pub fn register(&mut self, raw: *mut ffi::avn_world) -> ffi::avn_status {
        unsafe { ffi::avn_collision_hooks_set(raw, Some(filter_trampoline::<H>), Some(modify_trampoline::<H>), self as *mut Self as *mut c_void) }
    }
    pub fn unregister(raw: *mut ffi::avn_world) -> ffi::avn_status {
        unsafe { ffi::avn_collision_hooks_set(raw, None, None, std::ptr::null_mut()) }
    }
    fn collider(&self, index: u32) -> Entity { self.staging.collider_entities[self.staging.collider_slot[&index]] }
}

/// `avn_filter_pairs_fn`: `CollisionHooks::filter_pairs(collider1, collider2, commands)` per candidate pair; `should_collide` is preset to 1.
pub unsafe extern "C" fn filter_trampoline<H: CollisionHooks + 'static>(user: *mut c_void, n: u32, pairs: *const ffi::avn_hook_pair, should_collide: *mut u8)
where
    for<'x, 'y> SystemParamItem<'x, 'y, H>: CollisionHooks,
{
    let ctx = &mut *(user as *mut HookContext<'_, '_, '_, H>);
    let pairs = std::slice::from_raw_parts(pairs, n as usize);
    let keep = std::slice::from_raw_parts_mut(should_collide, n as usize);
    for (pair, keep) in pairs.iter().zip(keep.iter_mut()) {
        let (c1, c2) = (ctx.collider(pair.collider1), ctx.collider(pair.collider2));
        *keep = ctx.hooks.filter_pairs(c1, c2, ctx.commands) as u8;
    }
}

/// `avn_modify_contacts_fn`: the record -> a `ContactPair` with its one manifold (convex pairs) -> `CollisionHooks::modify_contacts` -> the record.
/// `touching` carries the hook's return value; an emptied `manifolds` comes back as `manifold_count = 0`; at most four points go back.
pub unsafe extern "C" fn modify_trampoline<H: CollisionHooks + 'static>(user: *mut c_void, scalar_bits: u32, n: u32, contacts: *mut c_void)
where
    for<'x, 'y> SystemParamItem<'x, 'y, H>: CollisionHooks,
{
    assert_eq!(scalar_bits, 32, "avian3d is built with the f32 feature in this crate");
    let ctx = &mut *(user as *mut HookContext<'_, '_, '_, H>);
    let records = std::slice::from_raw_parts_mut(contacts as *mut ffi::avn_hook_contact_f32, n as usize);
    let v3 = |a: &[f32], k: usize| Vec3::new(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
    for r in records.iter_mut() {
        let mut pair = ContactPair::new(ctx.collider(r.collider1), ctx.collider(r.collider2), ContactId(r.contact_id));
        pair.body1 = Some(ctx.staging.body_entities[r.body1 as usize]);
        pair.body2 = Some(ctx.staging.body_entities[r.body2 as usize]);
        pair.flags.set(ContactPairFlags::TOUCHING, r.flags & ffi::AVN_CP_TOUCHING != 0);   // (still the previous step's value when the hook runs, as in the reference)
        pair.flags.set(ContactPairFlags::GENERATE_CONSTRAINTS, r.flags & ffi::AVN_CP_GENERATE_CONSTRAINTS != 0);
        pair.flags.set(ContactPairFlags::STATIC1, r.flags & ffi::AVN_CP_STATIC1 != 0);
        pair.flags.set(ContactPairFlags::STATIC2, r.flags & ffi::AVN_CP_STATIC2 != 0);
        pair.flags.set(ContactPairFlags::MODIFY_CONTACTS, true);
        let points = (0..r.point_count as usize).map(|k| {
            // (world point: not kept by the device rows; anchor1 is relative to body 1's centre of mass -- a hook that needs the world point adds the body's position)
            let mut p = ContactPoint::new(v3(&r.anchor1, k), v3(&r.anchor2, k), Vec3::ZERO, r.penetration[k]);
            p.normal_speed = r.normal_speed[k];
            p.feature_id1 = PackedFeatureId(r.feature_id1[k]);
            p.feature_id2 = PackedFeatureId(r.feature_id2[k]);
            p
        });
        let mut manifold = ContactManifold::new(points, Vec3::from_array(r.normal));
        manifold.friction = r.friction;
        manifold.restitution = r.restitution;
        manifold.tangent_velocity = Vec3::from_array(r.tangent_velocity);
        pair.manifolds.push(manifold);

        let touching = ctx.hooks.modify_contacts(&mut pair, ctx.commands);

        r.touching = touching as u32;
        r.manifold_count = 0;
        r.point_count = 0;
        let Some(m) = pair.manifolds.first() else { continue };   // the hook emptied the list (or returned false: the device clears it either way)
        r.manifold_count = 1;
        r.normal = m.normal.to_array();
        r.friction = m.friction;
        r.restitution = m.restitution;
        r.tangent_velocity = m.tangent_velocity.to_array();
        for (k, p) in m.points.iter().take(ffi::AVN_MAX_MANIFOLD_POINTS as usize).enumerate() {
            r.anchor1[3 * k..3 * k + 3].copy_from_slice(&p.anchor1.to_array());
            r.anchor2[3 * k..3 * k + 3].copy_from_slice(&p.anchor2.to_array());
            r.penetration[k] = p.penetration;
            r.normal_speed[k] = p.normal_speed;
            r.feature_id1[k] = p.feature_id1.0;
            r.feature_id2[k] = p.feature_id2.0;
            r.point_count = k as u32 + 1;
        }
    }
}
