//! Golden vectors from REAL Avian (avianphysics/avian, the checkout this crate's `avian3d` path dependency names).
//!
//! `cargo run --release -- <out_dir> [scene ...]` runs each scene headless -- the app of the reference's own benches (`MinimalPlugins` + `TransformPlugin`
//! + `PhysicsPlugins::default()`, 60 Hz, `TimeUpdateStrategy::ManualDuration`, benches/src/dim3/mod.rs:28-49, benches/src/cli.rs:333-405) -- and writes
//! `<out_dir>/<scene>.avf`: a stream of chunks `[tag u32][frame u32][physics step u32][payload bytes u64][payload]`, little endian, every float as its BIT PATTERN:
//!
//!   META  one per file: substeps, dt bits, n bodies, n joints, gravity bits
//!   INIT  per body (spawn order = fixture id): rb type, shape (0 cuboid / 1 ball), half extents or radius, inverse mass, inverse inertia (m00 m01 m02 m11 m12 m22),
//!         centre of mass, friction, restitution  -- what avn_bodies_upload / avn_colliders_upload take, read from Avian's own components after the first update
//!   JNTS  per joint: type (AVN_JOINT_*), fixture ids of the bodies, local anchors, axis, limits, compliance
//!   BODY  per frame, per body: Position, Rotation (x y z w), LinearVelocity, AngularVelocity, SleepTimer, flags (bit 0 Sleeping)
//!   PAIR  per frame: the contact pairs that are NEW in this frame in ascending ContactId = the broad phase's emission order (IdPool hands out the lowest free id,
//!         data_structures/id_pool.rs:31-40): (contact id, fixture id of collider1, of collider2)
//!   LIVE  per frame: every live contact pair (contact id, flags, manifold count) in ascending id
//!   COLR  per frame: for each of the 24 GraphColors its `manifold_handles` IN ORDER (contact id, manifold index) -- constraint_graph.rs:66-97
//!   IMPL  per frame: per live touching pair and manifold point: contact id, manifold, point, feature ids, normal_impulse, warm_start_normal_impulse,
//!         warm_start_tangent_impulse, penetration, normal -- ContactPoint, contact_types/mod.rs:603-660
//!   ISLD  per frame, per body: IslandId of its BodyIslandNode (0xFFFFFFFF: none) -- islands/mod.rs:1314-1317
//!
//! tools/avian_fixtures_to_npz.py turns a file into tests/golden/avian/<scene>.npz; tests/test_reference_fixtures.py holds the oracle and the HIP path to it.
//! This crate cannot be built where it was written (no Rust toolchain in that image): tests/test_reference_fixtures.py checks every avian3d / bevy name it
//! uses against the reference's source instead.  Scenes: `cfg1` (BASELINE.json configs[0]: 10 x 10 x 10 unit cuboids, spacing 1.5, over a 200 x 1 x 200 slab,
//! 1 substep), `large_pyramid` (benches/src/dim3/large_pyramid.rs:15-40, base 20), `many_pyramids` (many_pyramids.rs:15-64, base 5, 3 x 3), `joints` (one
//! joint of each of the five types hanging from static anchors over a slab), `sleeping` (a 3 x 3 x 3 stack that settles and falls asleep, then a dropped box wakes it).
use std::{collections::HashMap, fs::File, io::{BufWriter, Write}, time::Duration};

use avian3d::{
    collision::contact_types::ContactId,
    dynamics::solver::{constraint_graph::ConstraintGraph, islands::BodyIslandNode},
    math::{Scalar, Vector},
    prelude::*,
};
use bevy::{prelude::*, time::TimeUpdateStrategy};

/// spawn order of a body = its index in every per-body array of the fixture (and the entity index the MI355X world is given for its collider)
#[derive(Component, Clone, Copy)]
struct FixtureId(u32);

/// physics steps taken so far (a system in `PhysicsSchedule`: an `app.update()` that runs no fixed step -- the first one -- leaves it unchanged)
#[derive(Resource, Default)]
struct PhysicsSteps(u32);

#[derive(Resource, Default)]
struct SeenContacts(HashMap<u32, (u32, u32)>);

#[derive(Clone, Copy)]
struct JointRec { kind: u32, b1: u32, b2: u32, a1: Vector, a2: Vector, axis: Vector, lim: [Scalar; 2], compliance: Scalar }

#[derive(Resource, Default)]
struct JointRecs(Vec<JointRec>);

struct Scene { name: &'static str, substeps: u32, frames: u32, setup: fn(&mut Commands, &mut JointRecs), after: Option<fn(&mut App, u32)> }

fn bits(x: Scalar) -> u32 { (x as f32).to_bits() }

struct Out { w: BufWriter<File> }
impl Out {
    fn chunk(&mut self, tag: &[u8; 4], frame: u32, step: u32, payload: &[u32]) {
        self.w.write_all(tag).unwrap();
        self.w.write_all(&frame.to_le_bytes()).unwrap();
        self.w.write_all(&step.to_le_bytes()).unwrap();
        self.w.write_all(&((payload.len() * 4) as u64).to_le_bytes()).unwrap();
        for v in payload { self.w.write_all(&v.to_le_bytes()).unwrap(); }
    }
}

fn spawn_box(commands: &mut Commands, id: &mut u32, rb: RigidBody, half: Vector, at: Vector) -> Entity {
    let e = commands.spawn((rb, Collider::cuboid(2.0 * half.x, 2.0 * half.y, 2.0 * half.z), Transform::from_xyz(at.x as f32, at.y as f32, at.z as f32), FixtureId(*id))).id();
    *id += 1;
    e
}

// BASELINE.json configs[0] as BASELINE.md section 3 defines it: a 200 x 1 x 200 static slab, 10 x 10 x 10 unit cuboids on a 1.5 grid starting 2 m above it
fn setup_cfg1(commands: &mut Commands, _j: &mut JointRecs) {
    let mut id = 0;
    spawn_box(commands, &mut id, RigidBody::Static, Vector::new(100.0, 0.5, 100.0), Vector::new(0.0, -0.5, 0.0));
    for y in 0..10 { for z in 0..10 { for x in 0..10 {
        let p = Vector::new((x as Scalar - 4.5) * 1.5, 2.0 + 0.5 + y as Scalar * 1.5, (z as Scalar - 4.5) * 1.5);
        spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), p);
    } } }
}
// benches/src/dim3/large_pyramid.rs:15-40 with base_count = 20 (210 boxes); the arithmetic is the bench's, in f32
fn setup_large_pyramid(commands: &mut Commands, _j: &mut JointRecs) {
    let base_count = 20usize;
    let mut id = 0;
    spawn_box(commands, &mut id, RigidBody::Static, Vector::new(400.0, 20.0, 400.0), Vector::new(0.0, -20.0, 0.0));
    let h = 0.5f32;
    let shift = h;
    for i in 0..base_count {
        let y = (2.0 * i as f32 + 1.0) * shift * 0.99;
        for j in i..base_count {
            let x = (i as f32 + 1.0) * shift + 2.0 * (j - i) as f32 * shift - h * base_count as f32;
            spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), Vector::new(x as Scalar, y as Scalar, 0.0));
        }
    }
}
// benches/src/dim3/many_pyramids.rs:15-64 with (base 5, 3 rows, 3 columns): 3 grounds, 9 pyramids of 15 boxes
fn setup_many_pyramids(commands: &mut Commands, _j: &mut JointRecs) {
    let (base_count, row_count, column_count) = (5usize, 3usize, 3usize);
    let mut id = 0;
    let h = 0.5f32;
    let ground_delta_y = 2.0 * h * (base_count + 1) as f32;
    let ground_width = 2.0 * h * column_count as f32 * (base_count + 1) as f32;
    for i in 0..row_count {
        spawn_box(commands, &mut id, RigidBody::Static, Vector::new((ground_width / 2.0) as Scalar, 0.005, (ground_width / 2.0) as Scalar), Vector::new(0.0, (i as f32 * ground_delta_y) as Scalar, 0.0));
    }
    let base_width = 2.0 * h * base_count as f32;
    for i in 0..row_count {
        let base_y = i as f32 * ground_delta_y;
        for j in 0..column_count {
            let center_x = -ground_width / 2.0 + j as f32 * (base_width + 2.0 * h) + h;
            for k in 0..base_count {
                let y = (2 * k + 1) as f32 * h + base_y;
                for l in k..base_count {
                    let x = (k + 1) as f32 * h + 2.0 * (l - k) as f32 * h + center_x - 0.5;
                    spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), Vector::new(x as Scalar, y as Scalar, 0.0));
                }
            }
        }
    }
}
// one joint of each type between a static anchor and a dynamic box (joint type numbers: include/avian_mi355x.h AVN_JOINT_*)
fn setup_joints(commands: &mut Commands, joints: &mut JointRecs) {
    let mut id = 0;
    spawn_box(commands, &mut id, RigidBody::Static, Vector::new(50.0, 0.5, 50.0), Vector::new(0.0, -0.5, 0.0));
    for k in 0..5u32 {
        let x = (k as Scalar - 2.0) * 6.0;
        let a = spawn_box(commands, &mut id, RigidBody::Static, Vector::splat(0.25), Vector::new(x, 8.0, 0.0));
        let b = spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), Vector::new(x + 2.0, 8.0, 0.0));
        let (ia, ib) = (id - 2, id - 1);
        let (a1, a2) = (Vector::ZERO, Vector::new(-2.0, 0.0, 0.0));
        let avn_kind = [4u32, 0, 1, 2, 3][k as usize];   // include/avian_mi355x.h: AVN_JOINT_FIXED 0, REVOLUTE 1, SPHERICAL 2, PRISMATIC 3, DISTANCE 4
        let mut rec = JointRec { kind: avn_kind, b1: ia, b2: ib, a1, a2, axis: Vector::Z, lim: [0.0, 0.0], compliance: 0.0 };
        match k {
            0 => { commands.spawn((DistanceJoint::new(a, b).with_local_anchor2(Vector::new(-0.5, 0.0, 0.0)).with_limits(1.5, 1.5), JointCollisionDisabled));
                   rec.a2 = Vector::new(-0.5, 0.0, 0.0); rec.lim = [1.5, 1.5]; }
            1 => { commands.spawn((FixedJoint::new(a, b).with_local_anchor2(a2), JointCollisionDisabled)); }
            2 => { commands.spawn((RevoluteJoint::new(a, b).with_local_anchor2(a2).with_hinge_axis(Vector::Z), JointCollisionDisabled)); }
            3 => { commands.spawn((SphericalJoint::new(a, b).with_local_anchor2(a2), JointCollisionDisabled)); }
            _ => { commands.spawn((PrismaticJoint::new(a, b).with_local_anchor2(a2).with_slider_axis(Vector::X).with_limits(-1.0, 1.0), JointCollisionDisabled));
                   rec.axis = Vector::X; rec.lim = [-1.0, 1.0]; }
        }
        joints.0.push(rec);
    }
}
// a 3 x 3 x 3 stack that comes to rest and falls asleep (TimeToSleep 0.5 s, the default thresholds); frame 150: a box dropped from 6 m wakes it
fn setup_sleeping(commands: &mut Commands, _j: &mut JointRecs) {
    let mut id = 0;
    spawn_box(commands, &mut id, RigidBody::Static, Vector::new(50.0, 0.5, 50.0), Vector::new(0.0, -0.5, 0.0));
    for y in 0..3 { for z in 0..3 { for x in 0..3 {
        spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), Vector::new(x as Scalar - 1.0, 0.5 + y as Scalar, z as Scalar - 1.0));
    } } }
    // the projectile exists from the start (a spawn in mid-run would move entity indices), parked far away and moved in by `after`
    spawn_box(commands, &mut id, RigidBody::Dynamic, Vector::splat(0.5), Vector::new(40.0, 0.5, 40.0));
}
fn sleeping_after(app: &mut App, frame: u32) {
    if frame != 150 { return; }
    let world = app.world_mut();
    let mut q = world.query::<(Entity, &FixtureId)>();
    let target = q.iter(world).find(|(_, f)| f.0 == 28).map(|(e, _)| e).unwrap();
    world.entity_mut(target).insert((Position(Vector::new(0.1, 9.0, 0.05)), LinearVelocity(Vector::ZERO)));
    world.commands().queue(WakeBody(target));
    world.flush();
}

const SCENES: &[Scene] = &[
    Scene { name: "cfg1", substeps: 1, frames: 120, setup: setup_cfg1, after: None },
    Scene { name: "large_pyramid", substeps: 4, frames: 90, setup: setup_large_pyramid, after: None },
    Scene { name: "many_pyramids", substeps: 4, frames: 90, setup: setup_many_pyramids, after: None },
    Scene { name: "joints", substeps: 4, frames: 120, setup: setup_joints, after: None },
    Scene { name: "sleeping", substeps: 4, frames: 260, setup: setup_sleeping, after: Some(sleeping_after) },
];

fn count_step(mut steps: ResMut<PhysicsSteps>) { steps.0 += 1; }

fn run(scene: &Scene, out_dir: &str) {
    let mut app = App::new();
    app.add_plugins((MinimalPlugins, TransformPlugin, PhysicsPlugins::default()));
    app.insert_resource(Time::<Fixed>::from_hz(60.0));
    app.insert_resource(TimeUpdateStrategy::ManualDuration(Duration::from_secs_f64(1.0 / 60.0)));
    app.insert_resource(SubstepCount(scene.substeps));
    app.init_resource::<PhysicsSteps>().init_resource::<SeenContacts>().init_resource::<JointRecs>();
    app.add_systems(PhysicsSchedule, count_step.in_set(PhysicsStepSystems::First));
    let setup = scene.setup;
    app.add_systems(Startup, move |mut commands: Commands, mut joints: ResMut<JointRecs>| setup(&mut commands, &mut joints));
    app.finish();
    app.cleanup();
    let mut out = Out { w: BufWriter::new(File::create(format!("{out_dir}/{}.avf", scene.name)).unwrap()) };
    let mut wrote_init = false;
    for frame in 0..scene.frames {
        if let Some(after) = scene.after { after(&mut app, frame); }
        app.update();
        let step = app.world().resource::<PhysicsSteps>().0;
        let world = app.world_mut();
        // bodies in fixture order
        let mut q = world.query::<(Entity, &FixtureId)>();
        let mut ents: Vec<(u32, Entity)> = q.iter(world).map(|(e, f)| (f.0, e)).collect();
        ents.sort();
        let fid: HashMap<Entity, u32> = ents.iter().map(|(f, e)| (*e, *f)).collect();
        if !wrote_init {
            wrote_init = true;
            let g = world.resource::<Gravity>().0;
            out.chunk(b"META", frame, step, &[scene.substeps, (1.0f32 / 60.0).to_bits(), ents.len() as u32, world.resource::<JointRecs>().0.len() as u32, bits(g.x), bits(g.y), bits(g.z)]);
            let mut p = Vec::new();
            for (_, e) in &ents {
                let er = world.entity(*e);
                let rb = match er.get::<RigidBody>().unwrap() { RigidBody::Dynamic => 0u32, RigidBody::Static => 1, RigidBody::Kinematic => 2 };
                let col = er.get::<Collider>().unwrap();
                let (shape, he) = if let Some(c) = col.shape().as_cuboid() { (0u32, [c.half_extents.x, c.half_extents.y, c.half_extents.z]) }
                                  else if let Some(b) = col.shape().as_ball() { (1u32, [b.radius, b.radius, b.radius]) } else { (99u32, [0.0, 0.0, 0.0]) };
                let im = er.get::<ComputedMass>().map(|m| m.inverse()).unwrap_or(0.0);
                let ii = er.get::<ComputedAngularInertia>().map(|i| i.inverse()).unwrap_or_default();
                let com = er.get::<ComputedCenterOfMass>().map(|c| c.0).unwrap_or(Vector::ZERO);
                let fr = er.get::<Friction>().copied().unwrap_or_default();
                let re = er.get::<Restitution>().copied().unwrap_or_default();
                p.extend([rb, shape, bits(he[0] as Scalar), bits(he[1] as Scalar), bits(he[2] as Scalar), bits(im), bits(ii.m00), bits(ii.m01), bits(ii.m02), bits(ii.m11), bits(ii.m12), bits(ii.m22),
                          bits(com.x), bits(com.y), bits(com.z), bits(fr.dynamic_coefficient), bits(re.coefficient)]);
            }
            out.chunk(b"INIT", frame, step, &p);
            let mut p = Vec::new();
            for j in &world.resource::<JointRecs>().0 {
                p.extend([j.kind, j.b1, j.b2, bits(j.a1.x), bits(j.a1.y), bits(j.a1.z), bits(j.a2.x), bits(j.a2.y), bits(j.a2.z), bits(j.axis.x), bits(j.axis.y), bits(j.axis.z),
                          bits(j.lim[0]), bits(j.lim[1]), bits(j.compliance)]);
            }
            out.chunk(b"JNTS", frame, step, &p);
        }
        let mut body = Vec::new();
        let mut isld = Vec::new();
        for (_, e) in &ents {
            let er = world.entity(*e);
            let (p, r) = (er.get::<Position>().unwrap().0, er.get::<Rotation>().unwrap().0);
            let v = er.get::<LinearVelocity>().map(|v| v.0).unwrap_or(Vector::ZERO);
            let w = er.get::<AngularVelocity>().map(|v| v.0).unwrap_or(Vector::ZERO);
            let t = er.get::<SleepTimer>().map(|t| t.0).unwrap_or(0.0);
            body.extend([bits(p.x), bits(p.y), bits(p.z), bits(r.x), bits(r.y), bits(r.z), bits(r.w), bits(v.x), bits(v.y), bits(v.z), bits(w.x), bits(w.y), bits(w.z),
                         t.to_bits(), er.contains::<Sleeping>() as u32]);
            isld.push(er.get::<BodyIslandNode>().map(|n| n.island_id().0).unwrap_or(u32::MAX));
        }
        out.chunk(b"BODY", frame, step, &body);
        out.chunk(b"ISLD", frame, step, &isld);
        // contact pairs: active and sleeping, by ascending ContactId
        let graph = world.resource::<ContactGraph>();
        let mut pairs: Vec<&ContactPair> = graph.iter_active().chain(graph.iter_sleeping()).collect();
        pairs.sort_by_key(|p| p.contact_id.0);
        let mut live = Vec::new();
        let mut imp = Vec::new();
        let mut now: HashMap<u32, (u32, u32)> = HashMap::new();
        for p in &pairs {
            let ContactId(id) = p.contact_id;
            let (c1, c2) = (fid.get(&p.collider1).copied().unwrap_or(u32::MAX), fid.get(&p.collider2).copied().unwrap_or(u32::MAX));
            now.insert(id, (c1, c2));
            live.extend([id, p.flags.bits() as u32, p.manifolds.len() as u32]);
            for (mi, m) in p.manifolds.iter().enumerate() {
                for (pi, pt) in m.points.iter().enumerate() {
                    imp.extend([id, mi as u32, pi as u32, pt.feature_id1.0, pt.feature_id2.0, bits(pt.normal_impulse), bits(pt.warm_start_normal_impulse),
                                bits(pt.warm_start_tangent_impulse.x), bits(pt.warm_start_tangent_impulse.y), bits(pt.penetration), bits(m.normal.x), bits(m.normal.y), bits(m.normal.z)]);
                }
            }
        }
        let mut new_pairs = Vec::new();
        {
            let seen = &world.resource::<SeenContacts>().0;
            let mut ids: Vec<&u32> = now.keys().collect();
            ids.sort();
            for id in ids { if seen.get(id) != now.get(id) { let (c1, c2) = now[id]; new_pairs.extend([*id, c1, c2]); } }
        }
        let mut colr = Vec::new();
        for color in &world.resource::<ConstraintGraph>().colors {
            colr.push(color.manifold_handles.len() as u32);
            for h in &color.manifold_handles { colr.extend([h.contact_id.0, h.manifold_index as u32]); }
        }
        out.chunk(b"PAIR", frame, step, &new_pairs);
        out.chunk(b"LIVE", frame, step, &live);
        out.chunk(b"COLR", frame, step, &colr);
        out.chunk(b"IMPL", frame, step, &imp);
        world.resource_mut::<SeenContacts>().0 = now;
    }
    out.w.flush().unwrap();
    println!("{}: {} frames, {} physics steps -> {out_dir}/{}.avf", scene.name, scene.frames, app.world().resource::<PhysicsSteps>().0, scene.name);
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() < 2 { eprintln!("usage: avian_fixtures <out_dir> [scene ...]   scenes: cfg1 large_pyramid many_pyramids joints sleeping"); std::process::exit(2); }
    std::fs::create_dir_all(&args[1]).unwrap();
    for scene in SCENES {
        if args.len() > 2 && !args[2..].iter().any(|a| a == scene.name) { continue; }
        run(scene, &args[1]);
    }
}
