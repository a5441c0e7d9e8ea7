//! MI355X plugins for Avian's 3D hot path.
//!
//! ```ignore
//! app.add_plugins(
//!     PhysicsPlugins::default()
//!         .build()
//!         .disable::<BroadPhasePlugin>()      // src/collision/broad_phase.rs:33-170
//!         // NarrowPhasePlugin stays: with the default `Mi355xMode::Auto` the step runs closed-loop on the device (Ball / Cuboid in kernels, other shapes through
//!         // host_shapes.rs, `CollisionHooks` through hooks.rs; Avian's narrow phase then walks an empty pair list) and falls back to Avian's own narrow phase only
//!         // for colliders that sit on no known rigid body; disable it only with an explicit `Mi355xMode::ClosedLoop`
//!         .disable::<IntegratorPlugin>()      // src/dynamics/integrator/mod.rs:45-88
//!         .disable::<SolverPlugin>()          // src/dynamics/solver/plugin.rs:88-151
//!         .disable::<XpbdSolverPlugin>()      // src/dynamics/solver/xpbd/plugin.rs:21-110: joints.rs uploads all five joint types with JointDamping /
//!                                             // JointCollisionDisabled and writes JointForces back
//!         .add(Mi355xPhysicsPlugin::<()>::default()),   // or ::<MyHooks>: the same `CollisionHooks` parameter as `PhysicsPlugins::with_collision_hooks::<MyHooks>()`
//! );
//! ```
//! `SolverSchedulePlugin` stays enabled: it owns the ordering of `SolverSystems` and the substep runner
//! (src/dynamics/solver/schedule.rs:32-49); the systems of this crate are added INTO the reference's own sets
//! (the mechanism of crates/avian3d/examples/custom_broad_phase.rs:63-72), so everything else in Avian keeps its place.
//!
//! Layers: [`world::Mi355xWorld`] is the safe owner of the `avn_world*`; [`staging::Staging`] turns ECS queries into the
//! Structure-of-Arrays the C ABI borrows for the duration of a call; [`plugins`] holds the systems of the rigid-body path, [`joints`] the
//! XpbdSolverPlugin replacement, [`closed_loop`] what the device closed loop owes the ECS (collision events, `CollidingEntities`, `Sleeping`),
//! [`host_shapes`] the two `AnyCollider` methods of every collider that is not a Ball / Cuboid, called back by the library (`avn_host_shapes_set`),
//! [`hooks`] the application's `CollisionHooks` (`filter_pairs`, `modify_contacts`), called back for the pairs of `ActiveCollisionHooks` colliders (`avn_collision_hooks_set`).
//!
//! Every component the recipe above takes away from Avian has a system here:
//!
//! | disabled plugin | what it did | here |
//! |---|---|---|
//! | BroadPhasePlugin | `AabbIntervals`, sweep-and-prune, new `ContactEdge`s (honouring `JointCollisionDisabled`) | `plugins::gpu_upload_bodies`, `gpu_broad_phase`; the joint-disabled pairs travel with `joints::gpu_upload_joints` |
//! | IntegratorPlugin | velocity increments, integrate velocities / positions, speed clamps | inside `avn_step` / `AVN_SYS_SOLVER` (`plugins::gpu_solver`) |
//! | SolverPlugin | constraint generation, warm start, solve / relax, restitution, impulse store, `joint_damping` | `plugins::gpu_upload_constraints`, `gpu_solver`, `gpu_download` |
//! | XpbdSolverPlugin | prepare / solve of Fixed, Revolute, Spherical, Prismatic, Distance joints, velocity projection, `JointForces` | `joints::gpu_upload_joints`, `gpu_solver`, `joints::gpu_download_joints` |
//! | (closed loop only) NarrowPhase's status loop | `CollisionStart` / `CollisionEnd`, `CollidingEntities` | `closed_loop::gpu_closed_loop_events` |
//! | (closed loop only) island sleeping | `Sleeping`, `SleepTimer`, wake on change | `closed_loop::gpu_closed_loop_sleeping`, `gpu_closed_loop_wake_on_changed` (the library's island manager decides: `avn_sleeping_enable`) |

pub mod closed_loop;
pub mod hooks;
pub mod host_shapes;
pub mod joints;
pub mod plugins;
pub mod staging;
pub mod world;

pub use plugins::{Mi355xMode, Mi355xPhysicsPlugin};
pub use world::{Mi355xError, Mi355xWorld};
