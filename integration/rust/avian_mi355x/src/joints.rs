//! XpbdSolverPlugin's side of the replacement: the five XPBD joint types, `JointDamping`, `JointCollisionDisabled` and `JointForces`.
//!
//! What the reference does per step (src/dynamics/solver/xpbd/plugin.rs): `prepare_xpbd_joint::<T>` for every joint type in
//! `SolverSystems::PrepareJoints` (:125-142: clear the Lagrange multipliers, `prepare` from the bodies' `Position` / `Rotation` /
//! `ComputedCenterOfMass`), `solve_xpbd_joint::<T>` per substep in the order Fixed, Revolute, Spherical, Prismatic, Distance (:77-82, :145-189),
//! `joint_damping::<T>` (src/dynamics/solver/plugin.rs:759-806), `writeback_joint_forces::<T>` (:242-260).  Here: one upload of the joint set
//! (`avn_joints_upload`; the library runs `prepare`, the per-type serial order and the damping itself, bit-identical to the serial loops) in
//! `SolverSystems::PrepareJoints`, one download of forces and torques (`avn_joints_download`) in `SolverSystems::Writeback`.
//!
//! Joint order: the library solves type by type and, inside a type, in ARRAY order; the reference iterates each type's query.  The staging keeps
//! uploaded joints in place and appends new ones by `Entity`, which is the order Bevy's dense table iteration yields for joints spawned in sequence.

use crate::{staging::Staging, world::Mi355xWorld};
use avian3d::prelude::*;
use avian_mi355x_sys as ffi;
use bevy::prelude::*;
use core::ffi::c_void;

/// `avn_joints` arrays + the entity of every uploaded joint (download target), reused across steps.
#[derive(Resource, Default)]
pub struct JointStaging {
    pub entities: Vec<Entity>,
    pub joint_type: Vec<u8>, pub body1: Vec<i32>, pub body2: Vec<i32>,
    pub local_anchor1: Vec<f32>, pub local_anchor2: Vec<f32>, pub local_basis1: Vec<f32>, pub local_basis2: Vec<f32>, pub axis: Vec<f32>,
    pub limit_min: Vec<f32>, pub limit_max: Vec<f32>, pub limit2_min: Vec<f32>, pub limit2_max: Vec<f32>, pub limit_flags: Vec<u8>,
    pub compliance: Vec<f32>, pub damping_linear: Vec<f32>, pub damping_angular: Vec<f32>, pub collision_disabled: Vec<u8>,
    pub out_force: Vec<f32>, pub out_torque: Vec<f32>, pub out_scratch: Vec<f32>,
    /// joints skipped this step: a body the staging does not know (despawned, or not a rigid body) or a frame still in `FromGlobal` form
    /// (Avian's own `JointPlugin` converts those before the solver runs; the reference's `prepare` returns early for them too, distance.rs:44-49)
    pub skipped: usize,
    pub any_damping: bool,
}

fn p<T>(v: &[T]) -> *const c_void { if v.is_empty() { core::ptr::null() } else { v.as_ptr() as *const c_void } }
fn push3(v: &mut Vec<f32>, a: Vec3) { v.extend_from_slice(&[a.x, a.y, a.z]); }
fn push4(v: &mut Vec<f32>, q: Quat) { v.extend_from_slice(&q.to_array()); }

struct Row {
    entity: Entity, kind: u32, body1: Entity, body2: Entity, anchor1: JointAnchor, anchor2: JointAnchor, basis1: JointBasis, basis2: JointBasis,
    axis: Vec3, limit: Option<(f32, f32)>, limit2: Option<(f32, f32)>, compliance: [f32; 3], damping: Option<(f32, f32)>, collision_disabled: bool,
}

pub type JointExtras<'a> = (Option<&'a JointDamping>, Has<JointCollisionDisabled>);
type Filter = (Without<RigidBody>, Without<JointDisabled>);   // the filter of prepare_xpbd_joint / solve_xpbd_joint (xpbd/plugin.rs:129, :150)

impl JointStaging {
    fn clear(&mut self) {
        macro_rules! clear { ($($f:ident),*) => { $( self.$f.clear(); )* } }
        clear!(entities, joint_type, body1, body2, local_anchor1, local_anchor2, local_basis1, local_basis2, axis, limit_min, limit_max, limit2_min, limit2_max,
               limit_flags, compliance, damping_linear, damping_angular, collision_disabled);
        self.skipped = 0;
        self.any_damping = false;
    }

    fn push(&mut self, st: &Staging, r: Row) {
        let (Some(&b1), Some(&b2)) = (st.body_index.get(&r.body1), st.body_index.get(&r.body2)) else { self.skipped += 1; return };
        let (JointAnchor::Local(a1), JointAnchor::Local(a2)) = (r.anchor1, r.anchor2) else { self.skipped += 1; return };
        let (JointBasis::Local(q1), JointBasis::Local(q2)) = (r.basis1, r.basis2) else { self.skipped += 1; return };
        self.entities.push(r.entity);
        self.joint_type.push(r.kind as u8); self.body1.push(b1); self.body2.push(b2);
        push3(&mut self.local_anchor1, a1); push3(&mut self.local_anchor2, a2); push4(&mut self.local_basis1, q1); push4(&mut self.local_basis2, q2);
        push3(&mut self.axis, r.axis);
        let (lo, hi) = r.limit.unwrap_or((0.0, 0.0)); self.limit_min.push(lo); self.limit_max.push(hi);
        let (lo2, hi2) = r.limit2.unwrap_or((0.0, 0.0)); self.limit2_min.push(lo2); self.limit2_max.push(hi2);
        self.limit_flags.push((if r.limit.is_some() { ffi::AVN_JOINT_HAS_LIMIT1 } else { 0 } | if r.limit2.is_some() { ffi::AVN_JOINT_HAS_LIMIT2 } else { 0 }) as u8);
        self.compliance.extend_from_slice(&r.compliance);
        // JointDamping is optional per joint and joint_damping::<T> only visits joints that carry it (solver/plugin.rs:761).  The ABI's damping columns
        // are all-or-nothing (NULL = no joint is damped): in a mixed set a joint without the component gets zero coefficients, i.e. `v += (v2 - v1) * 0`,
        // the same velocities up to the sign of a zero component; a set without any JointDamping uploads NULL columns (`any_damping`)
        if r.damping.is_some() { self.any_damping = true; }
        let (dl, da) = r.damping.unwrap_or((0.0, 0.0)); self.damping_linear.push(dl); self.damping_angular.push(da);
        self.collision_disabled.push(r.collision_disabled as u8);
    }

    /// The joint set of the step in the reference's solve order: type by type (xpbd/plugin.rs:77-82), each type's joints by `Entity`.
    #[allow(clippy::too_many_arguments)]
    pub fn fill<'a>(
        &mut self, st: &Staging,
        fixed: impl Iterator<Item = (Entity, &'a FixedJoint, JointExtras<'a>)>, revolute: impl Iterator<Item = (Entity, &'a RevoluteJoint, JointExtras<'a>)>,
        spherical: impl Iterator<Item = (Entity, &'a SphericalJoint, JointExtras<'a>)>, prismatic: impl Iterator<Item = (Entity, &'a PrismaticJoint, JointExtras<'a>)>,
        distance: impl Iterator<Item = (Entity, &'a DistanceJoint, JointExtras<'a>)>,
    ) {
        let damp = |d: Option<&JointDamping>| d.map(|d| (d.linear, d.angular));
        let mut rows: Vec<Row> = Vec::new();
        rows.extend(fixed.map(|(e, j, (d, cd))| Row {
            entity: e, kind: ffi::AVN_JOINT_FIXED, body1: j.body1, body2: j.body2, anchor1: j.frame1.anchor, anchor2: j.frame2.anchor, basis1: j.frame1.basis,
            basis2: j.frame2.basis, axis: Vec3::Z, limit: None, limit2: None, compliance: [j.point_compliance, j.angle_compliance, 0.0], damping: damp(d), collision_disabled: cd }));
        rows.extend(revolute.map(|(e, j, (d, cd))| Row {
            entity: e, kind: ffi::AVN_JOINT_REVOLUTE, body1: j.body1, body2: j.body2, anchor1: j.frame1.anchor, anchor2: j.frame2.anchor, basis1: j.frame1.basis,
            basis2: j.frame2.basis, axis: j.hinge_axis, limit: j.angle_limit.map(|l| (l.min, l.max)), limit2: None,
            compliance: [j.point_compliance, j.align_compliance, j.limit_compliance], damping: damp(d), collision_disabled: cd }));
        rows.extend(spherical.map(|(e, j, (d, cd))| Row {
            entity: e, kind: ffi::AVN_JOINT_SPHERICAL, body1: j.body1, body2: j.body2, anchor1: j.frame1.anchor, anchor2: j.frame2.anchor, basis1: j.frame1.basis,
            basis2: j.frame2.basis, axis: j.twist_axis, limit: j.swing_limit.map(|l| (l.min, l.max)), limit2: j.twist_limit.map(|l| (l.min, l.max)),
            compliance: [j.point_compliance, j.swing_compliance, j.twist_compliance], damping: damp(d), collision_disabled: cd }));
        rows.extend(prismatic.map(|(e, j, (d, cd))| Row {
            entity: e, kind: ffi::AVN_JOINT_PRISMATIC, body1: j.body1, body2: j.body2, anchor1: j.frame1.anchor, anchor2: j.frame2.anchor, basis1: j.frame1.basis,
            basis2: j.frame2.basis, axis: j.slider_axis, limit: j.limits.map(|l| (l.min, l.max)), limit2: None,
            compliance: [j.align_compliance, j.angle_compliance, j.limit_compliance], damping: damp(d), collision_disabled: cd }));
        rows.extend(distance.map(|(e, j, (d, cd))| Row {
            entity: e, kind: ffi::AVN_JOINT_DISTANCE, body1: j.body1, body2: j.body2, anchor1: j.anchor1, anchor2: j.anchor2, basis1: JointBasis::IDENTITY,
            basis2: JointBasis::IDENTITY, axis: Vec3::X, limit: Some((j.limits.min, j.limits.max)), limit2: None, compliance: [j.compliance, 0.0, 0.0],
            damping: damp(d), collision_disabled: cd }));
        // Array order: joints already uploaded keep their relative order, new ones are appended by `Entity` (spawn order).  The library solves type by
        // type whatever the array order is (a stable sort by type, xpbd/plugin.rs:77-82) and, inside a type, in array order = spawn order, like the
        // reference's per-type queries; and with closed-loop sleeping on it links joints into islands in the order it first sees them
        // (`PhysicsIslands::add_joint`, islands/mod.rs:668-735) and accepts only APPENDED joints between despawns.
        let previous: bevy::platform::collections::HashMap<Entity, usize> = self.entities.iter().enumerate().map(|(i, &e)| (e, i)).collect();
        rows.sort_by_key(|r| (previous.get(&r.entity).copied().unwrap_or(usize::MAX), r.entity));
        self.clear();
        for r in rows { self.push(st, r); }
    }

    pub fn desc(&self) -> ffi::avn_joints {
        ffi::avn_joints {
            count: self.entities.len() as u32, joint_type: self.joint_type.as_ptr(), body1: self.body1.as_ptr(), body2: self.body2.as_ptr(),
            local_anchor1: p(&self.local_anchor1), local_anchor2: p(&self.local_anchor2), local_basis1: p(&self.local_basis1), local_basis2: p(&self.local_basis2),
            axis: p(&self.axis), limit_min: p(&self.limit_min), limit_max: p(&self.limit_max), limit2_min: p(&self.limit2_min), limit2_max: p(&self.limit2_max),
            limit_flags: self.limit_flags.as_ptr(), compliance: p(&self.compliance),
            damping_linear: if self.any_damping { p(&self.damping_linear) } else { core::ptr::null() },
            damping_angular: if self.any_damping { p(&self.damping_angular) } else { core::ptr::null() },
            collision_disabled: self.collision_disabled.as_ptr(),
        }
    }

    /// `JointForces::force` / `::torque` of every uploaded joint (`writeback_joint_forces`, xpbd/plugin.rs:242-260); the solver-data fields of
    /// `avn_joints_out` land in one scratch buffer (inspection only, required non-NULL by the call).
    pub fn out_desc(&mut self) -> ffi::avn_joints_out {
        let n = self.entities.len();
        self.out_force.resize(3 * n, 0.0); self.out_torque.resize(3 * n, 0.0); self.out_scratch.resize(5 * 3 * n, 0.0);
        let s = self.out_scratch.as_mut_ptr();
        let at = |k: usize| if n == 0 { core::ptr::null_mut() } else { unsafe { s.add(k * 3 * n) as *mut c_void } };
        ffi::avn_joints_out {
            world_r1: at(0), world_r2: at(1), center_difference: at(2), total_lagrange: at(3), force: self.out_force.as_mut_ptr() as *mut c_void,
            total_rotation_lagrange: at(4), torque: self.out_torque.as_mut_ptr() as *mut c_void,
        }
    }
}

/// `SolverSystems::PrepareJoints`: the step's joint set -> `avn_joints_upload` (prepare_xpbd_joint of all five types happens on the device in
/// `AVN_SYS_SOLVER` / `avn_step`).  The `collision_disabled` column is what the broad phase's pair filter reads (`JointGraph` lookup of
/// src/collision/broad_phase.rs:421-429): the library builds its joint-disabled pair set from it.
#[allow(clippy::type_complexity)]
pub fn gpu_upload_joints(
    mut w: ResMut<Mi355xWorld>, st: Res<crate::plugins::Mi355xStaging>, mut js: ResMut<JointStaging>,
    fixed: Query<(Entity, &FixedJoint, JointExtras<'static>), Filter>, revolute: Query<(Entity, &RevoluteJoint, JointExtras<'static>), Filter>,
    spherical: Query<(Entity, &SphericalJoint, JointExtras<'static>), Filter>, prismatic: Query<(Entity, &PrismaticJoint, JointExtras<'static>), Filter>,
    distance: Query<(Entity, &DistanceJoint, JointExtras<'static>), Filter>,
) {
    let had = !js.entities.is_empty();
    js.fill(&st.0, fixed.iter(), revolute.iter(), spherical.iter(), prismatic.iter(), distance.iter());
    if js.entities.is_empty() && !had { return; }   // a world without joints never calls into the joint path
    let d = js.desc();
    let raw = w.raw();
    let s = unsafe { ffi::avn_joints_upload(raw, &d) };
    w.check(s);
}

/// `SolverSystems::Writeback`, next to the body write-back: `writeback_joint_forces::<T>` for all five types.
pub fn gpu_download_joints(mut w: ResMut<Mi355xWorld>, mut js: ResMut<JointStaging>, mut forces: Query<&mut JointForces>) {
    if js.entities.is_empty() { return; }
    let out = js.out_desc();
    let raw = w.raw();
    let s = unsafe { ffi::avn_joints_download(raw, &out) };
    w.check(s);
    for (i, &e) in js.entities.iter().enumerate() {
        let Ok(mut f) = forces.get_mut(e) else { continue };   // only joints whose entity carries `JointForces` are written (the query of :243)
        f.set_force(Vec3::from_slice(&js.out_force[3 * i..]));
        f.set_torque(Vec3::from_slice(&js.out_torque[3 * i..]));
    }
}
