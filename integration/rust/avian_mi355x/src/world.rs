//! Safe ownership of the device world.  One caller thread at a time, like the single-threaded executor of `PhysicsSchedule`
//! (src/schedule/mod.rs:90): the handle is `Send` (Bevy resources move between threads) but not `Sync`.

use avian_mi355x_sys as ffi;
use bevy::prelude::Resource;
use std::ffi::CStr;

#[derive(Debug)]
pub struct Mi355xError {
    pub status: ffi::avn_status,
    pub message: String,
}

impl core::fmt::Display for Mi355xError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "avian_mi355x: status {}: {}", self.status, self.message)
    }
}
impl std::error::Error for Mi355xError {}

/// The device-resident physics world (`avn_world*`).
#[derive(Resource)]
pub struct Mi355xWorld {
    raw: *mut ffi::avn_world,
    pub config: ffi::avn_config,
    closed_loop: bool,
}
// SAFETY: the library keeps no thread-local state per world; `&mut self` on every call serialises access.
unsafe impl Send for Mi355xWorld {}

impl Mi355xWorld {
    /// `avn_world_create`.  Fails with `AVN_ERR_NO_DEVICE` when no gfx950 device is visible: the path has no CPU fallback, the
    /// caller keeps the stock plugins in that case.
    pub fn new(config: ffi::avn_config) -> Result<Self, Mi355xError> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: `config` is a valid `avn_config` with `struct_size` set by `default_config`.
        let status = unsafe { ffi::avn_world_create(&config, &mut raw) };
        if status != ffi::AVN_OK {
            let message = unsafe { cstr(ffi::avn_last_error(core::ptr::null())) };
            return Err(Mi355xError { status, message });
        }
        Ok(Self { raw, config, closed_loop: false })
    }

    /// The reference's convention on invariant violations is to panic (src/dynamics/solver/plugin.rs:393-395, 735-739;
    /// src/collision/broad_phase.rs:469-471): systems call this on every status.
    #[track_caller]
    pub fn check(&self, status: ffi::avn_status) {
        if status != ffi::AVN_OK {
            let message = unsafe { cstr(ffi::avn_last_error(self.raw)) };
            panic!("avian_mi355x: status {status}: {message}");
        }
    }

    pub fn raw(&mut self) -> *mut ffi::avn_world {
        self.raw
    }

    /// Resources changed (`SubstepCount`, `Gravity`, `SolverConfig`, `PhysicsLengthUnit`, `NarrowPhaseConfig`, `Time<Physics>` delta).
    pub fn set_config(&mut self, config: ffi::avn_config) {
        self.config = config;
        let st = unsafe { ffi::avn_config_set(self.raw, &self.config) };
        self.check(st);
    }

    pub fn run_system(&mut self, system: ffi::avn_system) {
        let st = unsafe { ffi::avn_run_system(self.raw, system) };
        self.check(st);
    }

    pub fn step(&mut self) {
        let st = unsafe { ffi::avn_step(self.raw) };
        self.check(st);
    }

    /// `avn_pipeline_enable(1 / 0)` when the wanted state differs from the current one (the plugin's effective mode can change from one
    /// step to the next: a collider gained or lost `ActiveCollisionHooks`).  Leaving the loop drops the device contact rows; entering it
    /// starts from an empty ContactGraph on the device (include/avian_mi355x.h: avn_pipeline_enable).
    pub fn set_closed_loop(&mut self, on: bool) {
        if on == self.closed_loop { return; }
        let st = unsafe { ffi::avn_pipeline_enable(self.raw, if on { 1 } else { 0 }) };
        self.check(st);
        self.closed_loop = on;
    }

    /// Entities that left the world since the last step, inside the closed loop (`avn_despawn`): what Avian's `remove_body_on` /
    /// `remove_collider_on` observers do to its own ContactGraph / ConstraintGraph / PhysicsIslands (src/collision/narrow_phase/mod.rs:399-560)
    /// happens to the device's -- pops in edge-list order, ContactIds back into the IdPool, the bodies' island nodes unlinked -- and the library
    /// renumbers the remaining bodies by stable compaction.  The caller uploads the remaining bodies and colliders next (same call order as
    /// every frame: `avn_bodies_upload`, `avn_colliders_upload`).  A no-op outside the closed loop (the graphs are Avian's own there).
    /// `joints`: indices into the last `avn_joints_upload` of the joints that leave -- the joint entities that were despawned and every joint that names a
    /// despawned body (Avian keeps such a joint, dangling, and skips it in every system; the library holds none).  They leave first:
    /// `remove_joint_from_graph` (src/dynamics/solver/joint_graph/plugin.rs:163-194).
    pub fn despawn(&mut self, bodies: &[u32], collider_entities: &[u32], joints: &[u32]) {
        if !self.closed_loop || (bodies.is_empty() && collider_entities.is_empty() && joints.is_empty()) { return; }
        let d = ffi::avn_despawn_list {
            struct_size: core::mem::size_of::<ffi::avn_despawn_list>() as u32,
            n_colliders: collider_entities.len() as u32, collider_entities: if collider_entities.is_empty() { core::ptr::null() } else { collider_entities.as_ptr() },
            n_bodies: bodies.len() as u32, bodies: if bodies.is_empty() { core::ptr::null() } else { bodies.as_ptr() },
            n_joints: joints.len() as u32, joints: if joints.is_empty() { core::ptr::null() } else { joints.as_ptr() },
        };
        let st = unsafe { ffi::avn_despawn(self.raw, &d) };
        self.check(st);
    }

    /// Persistent islands + sleeping inside the closed loop (`avn_sleeping_enable`): the library keeps its own island manager and actuates
    /// sleeping on the device; `None` switches it off again.
    pub fn set_sleeping(&mut self, params: Option<ffi::avn_sleep_params>) {
        let st = unsafe { ffi::avn_sleeping_enable(self.raw, params.as_ref().map_or(core::ptr::null(), |p| p as *const _)) };
        self.check(st);
    }

    /// `WakeBody` for bodies the application moved or kicked (src/dynamics/solver/islands/sleeping.rs:556-604 wake_on_changed).
    pub fn wake_bodies(&mut self, bodies: &[u32]) {
        let st = unsafe { ffi::avn_wake_bodies(self.raw, bodies.as_ptr(), bodies.len()) };
        self.check(st);
    }

    pub fn sleeping_stats(&mut self) -> ffi::avn_sleeping_stats {
        let mut s = unsafe { core::mem::zeroed::<ffi::avn_sleeping_stats>() };
        let st = unsafe { ffi::avn_sleeping_stats_get(self.raw, &mut s) };
        self.check(st);
        s
    }

    /// New broad-phase pairs of the last `AVN_SYS_COLLECT_COLLISION_PAIRS`, in the reference's emission order.  The slice is owned by
    /// the library and valid until the next call on this world.
    pub fn pairs(&mut self) -> &[ffi::avn_pair] {
        let (mut p, mut n) = (core::ptr::null(), 0usize);
        let st = unsafe { ffi::avn_pairs_get(self.raw, &mut p, &mut n) };
        self.check(st);
        if n == 0 { &[] } else { unsafe { core::slice::from_raw_parts(p, n) } }
    }

    pub fn is_closed_loop(&self) -> bool { self.closed_loop }

    /// ContactIds the closed loop gave the last step's new pairs, entry i for pair i of [`Self::pairs`] (`avn_pipeline_new_pair_ids_get`).
    pub fn new_pair_ids(&mut self) -> &[u32] {
        let (mut p, mut n) = (core::ptr::null(), 0usize);
        let st = unsafe { ffi::avn_pipeline_new_pair_ids_get(self.raw, &mut p, &mut n) };
        self.check(st);
        if n == 0 { &[] } else { unsafe { core::slice::from_raw_parts(p, n) } }
    }

    /// Status changes of the last `AVN_SYS_NARROW_PHASE`, ascending `ContactId` (the order of the status-bit walk,
    /// src/collision/narrow_phase/system_param.rs:141-145).
    pub fn contact_changes(&mut self) -> &[ffi::avn_contact_change] {
        let (mut p, mut n) = (core::ptr::null(), 0usize);
        let st = unsafe { ffi::avn_contact_changes_get(self.raw, &mut p, &mut n) };
        self.check(st);
        if n == 0 { &[] } else { unsafe { core::slice::from_raw_parts(p, n) } }
    }

    /// `SolverDiagnostics` + `CollisionDiagnostics` of the last step (src/dynamics/solver/diagnostics.rs:13-37,
    /// src/collision/diagnostics.rs:13-19), milliseconds.
    pub fn diagnostics(&mut self) -> ffi::avn_diagnostics {
        let mut d = unsafe { core::mem::zeroed::<ffi::avn_diagnostics>() };
        let st = unsafe { ffi::avn_diagnostics_get(self.raw, &mut d) };
        self.check(st);
        d
    }
}

/// Islands, the sleeping decision and multi-GPU sharding (include/avian_mi355x.h, "islands and sleeping", "level-2 sharding").
impl Mi355xWorld {
    /// `update_sleeping_states` + the decision of `sleep_islands` for the step just taken (src/dynamics/solver/islands/sleeping.rs:184-280).
    /// The caller (the Sleeping set's system in `plugins.rs`) applies it: `SleepIslands` / `WakeIslands` stay host-side.
    pub fn sleep_update(&mut self, time_to_sleep: f32, threshold: (f32, f32), delta_secs: f32, length_unit: f64,
                        per_body: Option<(&[f32], &[f32], &[u8])>) -> ffi::avn_sleep_stats {
        let p = ffi::avn_sleep_params {
            struct_size: core::mem::size_of::<ffi::avn_sleep_params>() as u32,
            time_to_sleep,
            linear_threshold: threshold.0,
            angular_threshold: threshold.1,
            delta_secs,
            length_unit,
            // per-body `SleepThreshold` / `SleepingDisabled`: staged by the caller when any body overrides the defaults (plugins.rs)
            body_linear_threshold: per_body.map_or(core::ptr::null(), |p| p.0.as_ptr()),
            body_angular_threshold: per_body.map_or(core::ptr::null(), |p| p.1.as_ptr()),
            body_sleeping_disabled: per_body.map_or(core::ptr::null(), |p| p.2.as_ptr()),
        };
        let mut stats = unsafe { core::mem::zeroed::<ffi::avn_sleep_stats>() };
        let st = unsafe { ffi::avn_sleep_update(self.raw, &p, &mut stats) };
        self.check(st);
        stats
    }

    /// Per body: `SleepTimer`, island label (lowest body index), "island rests", "island wakes".
    pub fn sleep_state(&mut self, n_bodies: usize) -> (Vec<f32>, Vec<u32>, Vec<u8>, Vec<u8>) {
        let (mut t, mut l, mut r, mut k) = (vec![0f32; n_bodies], vec![0u32; n_bodies], vec![0u8; n_bodies], vec![0u8; n_bodies]);
        let out = ffi::avn_sleep_out { sleep_timer: t.as_mut_ptr(), island: l.as_mut_ptr(), island_rests: r.as_mut_ptr(), island_wakes: k.as_mut_ptr() };
        let st = unsafe { ffi::avn_sleep_get(self.raw, &out) };
        self.check(st);
        (t, l, r, k)
    }

    /// What `WakeIslands::apply` does to the timers of the bodies it wakes (sleeping.rs:492).
    pub fn sleep_reset(&mut self, bodies: &[u32]) {
        let st = unsafe { ffi::avn_sleep_reset(self.raw, bodies.as_ptr(), bodies.len()) };
        self.check(st);
    }

    /// Contact-table rows of pairs that move to another world (a re-partition hands an island to another rank): `rows_out` of the world
    /// that gives them up, `rows_in` on the one that takes them, after `avn_contact_pairs_add` created the rows there.  What travels is what
    /// `NarrowPhase::update` reads of the previous step: the `ContactPair` flags and `manifolds[0]` with feature ids and warm-start impulses.
    pub fn contact_rows_out(&mut self, contact_ids: &[u32]) -> ContactRows {
        let n = contact_ids.len();
        let mut r = ContactRows::zeroed(n);
        let out = ffi::avn_contacts_out {
            flags: r.flags.as_mut_ptr(), point_count: r.point_count.as_mut_ptr(), normal: r.normal.as_mut_ptr().cast(), friction: r.friction.as_mut_ptr().cast(),
            restitution: r.restitution.as_mut_ptr().cast(), anchor1: r.anchor1.as_mut_ptr().cast(), anchor2: r.anchor2.as_mut_ptr().cast(),
            penetration: r.penetration.as_mut_ptr().cast(), normal_speed: r.normal_speed.as_mut_ptr().cast(),
            warm_start_normal_impulse: r.warm_start_normal_impulse.as_mut_ptr().cast(), warm_start_tangent_impulse: r.warm_start_tangent_impulse.as_mut_ptr().cast(),
            normal_impulse: r.normal_impulse.as_mut_ptr().cast(), feature_id1: r.feature_id1.as_mut_ptr(), feature_id2: r.feature_id2.as_mut_ptr(),
        };
        let st = unsafe { ffi::avn_contacts_download(self.raw, contact_ids.as_ptr(), n, &out) };
        self.check(st);
        r
    }

    pub fn contact_rows_in(&mut self, contact_ids: &[u32], r: &ContactRows) {
        assert_eq!(r.flags.len(), contact_ids.len());
        let inp = ffi::avn_contacts_in {
            flags: r.flags.as_ptr(), point_count: r.point_count.as_ptr(), normal: r.normal.as_ptr().cast(), friction: r.friction.as_ptr().cast(),
            restitution: r.restitution.as_ptr().cast(), anchor1: r.anchor1.as_ptr().cast(), anchor2: r.anchor2.as_ptr().cast(),
            penetration: r.penetration.as_ptr().cast(), normal_speed: r.normal_speed.as_ptr().cast(),
            warm_start_normal_impulse: r.warm_start_normal_impulse.as_ptr().cast(), warm_start_tangent_impulse: r.warm_start_tangent_impulse.as_ptr().cast(),
            normal_impulse: r.normal_impulse.as_ptr().cast(), feature_id1: r.feature_id1.as_ptr(), feature_id2: r.feature_id2.as_ptr(),
        };
        let st = unsafe { ffi::avn_contacts_upload(self.raw, contact_ids.as_ptr(), contact_ids.len(), &inp) };
        self.check(st);
    }

    /// Level 2: this rank's send / receive lists (from `avn_level2_plan_rank`), then the library's own transport.
    pub fn enable_level2(&mut self, halo: &ffi::avn_halo_plan, unique_id: &[u8; ffi::AVN_COMM_ID_BYTES as usize], n_ranks: i32, rank: i32) {
        let st = unsafe { ffi::avn_halo_plan_upload(self.raw, halo) };
        self.check(st);
        let st = unsafe { ffi::avn_comm_init(self.raw, unique_id.as_ptr(), n_ranks, rank) };
        self.check(st);
    }
}

/// Rows of the contact table in the layout of `avn_contacts_out` / `avn_contacts_in` (f32 worlds; slot = 4 * row + point).
pub struct ContactRows {
    pub flags: Vec<u32>, pub point_count: Vec<u8>, pub normal: Vec<f32>, pub friction: Vec<f32>, pub restitution: Vec<f32>,
    pub anchor1: Vec<f32>, pub anchor2: Vec<f32>, pub penetration: Vec<f32>, pub normal_speed: Vec<f32>,
    pub warm_start_normal_impulse: Vec<f32>, pub warm_start_tangent_impulse: Vec<f32>, pub normal_impulse: Vec<f32>,
    pub feature_id1: Vec<u32>, pub feature_id2: Vec<u32>,
}
impl ContactRows {
    pub fn zeroed(n: usize) -> Self {
        Self {
            flags: vec![0; n], point_count: vec![0; n], normal: vec![0.0; 3 * n], friction: vec![0.0; n], restitution: vec![0.0; n],
            anchor1: vec![0.0; 12 * n], anchor2: vec![0.0; 12 * n], penetration: vec![0.0; 4 * n], normal_speed: vec![0.0; 4 * n],
            warm_start_normal_impulse: vec![0.0; 4 * n], warm_start_tangent_impulse: vec![0.0; 8 * n], normal_impulse: vec![0.0; 4 * n],
            feature_id1: vec![0; 4 * n], feature_id2: vec![0; 4 * n],
        }
    }
}

impl Drop for Mi355xWorld {
    fn drop(&mut self) {
        unsafe { ffi::avn_world_destroy(self.raw) }
    }
}

unsafe fn cstr(p: *const core::ffi::c_char) -> String {
    if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}

/// `avn_config` with the defaults of the reference's resources: `SubstepCount` 6 (src/dynamics/solver/schedule.rs:185-191),
/// `Gravity` (0, -9.81, 0), `SolverConfig::default()` (src/dynamics/solver/plugin.rs:216-302), `NarrowPhaseConfig::default()`
/// (src/collision/narrow_phase/mod.rs:203-255), `PhysicsLengthUnit` 1.
pub fn default_config() -> ffi::avn_config {
    ffi::avn_config {
        struct_size: core::mem::size_of::<ffi::avn_config>() as u32,
        scalar_bits: 32,
        device: 0,
        substeps: 6,
        dt_ns: 1_000_000_000 / 64, // Time<Fixed> default 64 Hz
        gravity: [0.0, -9.81, 0.0],
        length_unit: 1.0,
        contact_damping_ratio: 10.0,
        contact_frequency_factor: 1.5,
        max_overlap_solve_speed: 4.0,
        warm_start_coefficient: 1.0,
        restitution_threshold: 1.0,
        restitution_iterations: 1,
        match_contacts: 1,
        default_speculative_margin: f64::MAX,
        contact_tolerance: 0.005,
        solver_iterations: 1,
        use_graph: 1,
    }
}
