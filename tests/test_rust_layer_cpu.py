"""CPU: the hand-written Rust layer (integration/rust/avian_mi355x, uncompiled here: no Rust toolchain in the image) only names things that exist.
Every `ffi::avn_*` function, `ffi::avn_*` struct and `ffi::AVN_*` constant it uses must be declared by the generated bindings (avian_mi355x-sys, which
tests/test_abi_cpu.py checks against include/avian_mi355x.h), every struct literal must name exactly the fields of the C struct, and every system the
crate's documentation promises for a plugin the recipe disables must exist and be registered by `Mi355xPhysicsPlugin::build`."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "integration", "rust", "avian_mi355x", "src")
SYS = os.path.join(REPO, "integration", "rust", "avian_mi355x-sys", "src", "lib.rs")


def read(name):
    return open(os.path.join(SRC, name)).read()


def sources():
    return {n: read(n) for n in sorted(os.listdir(SRC)) if n.endswith(".rs")}


def test_every_ffi_name_the_layer_uses_is_in_the_generated_bindings():
    sys_src = open(SYS).read()
    fns = set(re.findall(r"pub fn (avn_\w+)\(", sys_src))
    structs = set(re.findall(r"pub struct (avn_\w+)", sys_src)) | set(re.findall(r"pub type (avn_\w+)", sys_src))
    consts = set(re.findall(r"pub const (AVN_\w+)", sys_src))
    assert len(fns) > 80 and len(structs) > 20 and len(consts) > 60
    used = set()
    for name, src in sources().items():
        for ident in re.findall(r"(?<![\w:])ffi::(\w+)", src):
            used.add(ident)
            assert ident in fns or ident in structs or ident in consts, f"{name}: ffi::{ident} is not declared by avian_mi355x-sys"
    for must in ("avn_joints_upload", "avn_joints_download", "avn_pipeline_new_pair_ids_get", "avn_contact_changes_get", "avn_sleeping_enable", "avn_sleeping_state_get",
                 "avn_wake_bodies", "avn_despawn", "avn_step", "avn_bodies_upload", "avn_colliders_upload", "avn_manifolds_upload", "avn_pipeline_enable"):
        assert must in used, f"the Rust layer never calls {must}"


def test_struct_literals_name_the_fields_of_the_c_structs():
    sys_src = open(SYS).read()
    fields = {}
    for m in re.finditer(r"pub struct (avn_\w+) \{(.*?)\n\}", sys_src, re.S):
        fields[m.group(1)] = set(re.findall(r"pub (\w+):", m.group(2)))
    checked = 0
    for name, src in sources().items():
        for m in re.finditer(r"(?<![\w:])ffi::(avn_\w+) \{", src):
            struct = m.group(1)
            if src[:m.start()].rstrip().endswith("->"): continue   # a return type in front of a function body, not a literal
            depth, i = 1, m.end()
            while depth and i < len(src):   # the literal's body, nested braces included
                depth += {"{": 1, "}": -1}.get(src[i], 0); i += 1
            body = re.sub(r"//[^\n]*", "", src[m.end():i - 1])
            flat, d = "", 0
            for ch in body:   # only the literal's own `field:` keys (depth 0)
                if ch in "({[": d += 1
                elif ch in ")}]": d -= 1
                flat += ch if d == 0 else " "
            keys = set(re.findall(r"(?:^|,)\s*(\w+)\s*(?=:|,|$)", flat))   # `field: value` and the `field,` shorthand
            if not keys:
                continue
            assert struct in fields, f"{name}: ffi::{struct} is not a struct of the bindings"
            assert keys == fields[struct], f"{name}: ffi::{struct} literal names {sorted(keys ^ fields[struct])} differently from the C struct"
            checked += 1
    assert checked >= 8


def test_every_promised_system_exists_and_is_registered():
    src = sources()
    everything = "\n".join(src.values())
    lib = src["lib.rs"]
    promised = set(re.findall(r"`(?:\w+::)?(gpu_\w+)`", lib))
    assert {"gpu_upload_joints", "gpu_download_joints", "gpu_closed_loop_events", "gpu_closed_loop_sleeping", "gpu_solver", "gpu_broad_phase"} <= promised
    build = src["plugins.rs"][src["plugins.rs"].index("fn build"):src["plugins.rs"].index("fn sync_config")]
    for system in promised:
        assert re.search(rf"fn {system}(<[^(]*>)?\(", everything), f"lib.rs promises `{system}`: no such function in the crate"
        assert system in build, f"`{system}` exists but Mi355xPhysicsPlugin::build never adds it to the schedule"
    # the recipe disables XpbdSolverPlugin: all five joint types, their damping, their collision switch and their forces must be staged
    joints = src["joints.rs"]
    for needed in ("FixedJoint", "RevoluteJoint", "SphericalJoint", "PrismaticJoint", "DistanceJoint", "JointDamping", "JointCollisionDisabled", "JointForces", "JointDisabled"):
        assert needed in joints, f"joints.rs never mentions {needed}"
    events = src["closed_loop.rs"]
    for needed in ("CollisionStart", "CollisionEnd", "CollidingEntities", "Sleeping", "SleepTimer", "SleepingDisabled"):
        assert needed in events, f"closed_loop.rs never mentions {needed}"
    # nothing in the crate may name a function that does not exist (round 4: plugins.rs promised a gpu_closed_loop_events that was never written)
    for name, s in src.items():
        for mentioned in set(re.findall(r"`(?:\w+::)*(gpu_\w+)`", s)):
            assert re.search(rf"fn {mentioned}(<[^(]*>)?\(", everything), f"{name} mentions `{mentioned}`, which does not exist"


def test_host_shape_trampolines_use_the_reference_names_and_the_header_records():
    """host_shapes.rs answers avn_host_aabb_fn / avn_host_manifolds_fn with the reference's own SimpleCollider methods and ContactPoint fields: every one of them
    must exist in /root/reference (when present here), the trampolines must be registered, and fill_colliders must route non-Ball / Cuboid shapes to the table."""
    src = sources()
    hs = src["host_shapes.rs"]
    for used in ("avn_host_shapes_set", "avn_host_aabb_query_f32", "avn_host_aabb_f32", "avn_host_manifold_query_f32", "avn_host_manifold_f32", "AVN_MAX_QUERY_POINTS"):
        assert f"ffi::{used}" in hs
    assert "AVN_SHAPE_HOST" in src["staging.rs"] and "host_shapes.colliders.insert" in src["staging.rs"]
    assert "host_shapes.register(raw)" in src["plugins.rs"] and "init_resource::<crate::host_shapes::HostShapeTable>" in src["plugins.rs"]
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        return
    collider_mod = open(os.path.join(ref, "collision", "collider", "mod.rs")).read()
    contact_types = open(os.path.join(ref, "collision", "contact_types", "mod.rs")).read()
    for method in ("fn aabb(", "fn swept_aabb(", "fn contact_manifolds("):   # SimpleCollider (collider/mod.rs:263-320)
        assert method in collider_mod and "." + method[3:] in hs, method
    for field in ("anchor1", "penetration", "feature_id1", "feature_id2"):
        assert re.search(rf"pub {field}:", contact_types) and f"p.{field}" in hs, field
    assert re.search(r"pub normal: Vector", contact_types) and "m.normal" in hs and "m.points" in hs
    assert re.search(r"pub min: Vector", collider_mod) and "aabb.min" in hs and "aabb.max" in hs


def test_hook_trampolines_use_the_reference_names_and_the_header_records():
    """hooks.rs answers avn_filter_pairs_fn / avn_modify_contacts_fn with the application's CollisionHooks: the trait methods, the ContactPair / ContactManifold /
    ContactPoint fields it reads and writes and the ActiveCollisionHooks bits the staging tests must exist in /root/reference (when present here); the context must be
    registered around avn_step by the (generic) solver system; a collider with hooks must no longer push the plugin out of the closed loop."""
    src = sources()
    hk = src["hooks.rs"]
    for used in ("avn_collision_hooks_set", "avn_hook_pair", "avn_hook_contact_f32", "AVN_MAX_MANIFOLD_POINTS", "AVN_CP_TOUCHING", "AVN_CP_GENERATE_CONSTRAINTS"):
        assert f"ffi::{used}" in hk
    plugins = src["plugins.rs"]
    assert "gpu_solver::<H>" in plugins and "HookContext::<H>" in plugins and "ctx.register(raw)" in plugins and "unregister(raw)" in plugins
    assert "StaticSystemParam<H>" in plugins and "SystemParamItem<'w, 's, H>: CollisionHooks" in plugins
    mode = plugins[plugins.index("fn effective_mode"):plugins.index("impl<H: CollisionHooks + 'static> Plugin")]
    assert "colliders_with_hooks" not in mode, "hooks no longer decide the mode: they are called back inside the closed loop"
    assert "ActiveCollisionHooks::FILTER_PAIRS" in src["staging.rs"] and "ActiveCollisionHooks::MODIFY_CONTACTS" in src["staging.rs"]
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        return
    hooks_rs = open(os.path.join(ref, "collision", "hooks.rs")).read()
    contact_types = open(os.path.join(ref, "collision", "contact_types", "mod.rs")).read()
    broad = open(os.path.join(ref, "collision", "broad_phase.rs")).read()
    assert "fn filter_pairs(&self, collider1: Entity, collider2: Entity, commands: &mut Commands) -> bool" in hooks_rs and ".filter_pairs(c1, c2, ctx.commands)" in hk
    assert "fn modify_contacts(&self, contacts: &mut ContactPair, commands: &mut Commands) -> bool" in hooks_rs and ".modify_contacts(&mut pair, ctx.commands)" in hk
    assert "const FILTER_PAIRS" in hooks_rs and "const MODIFY_CONTACTS" in hooks_rs
    assert "for<'w, 's> SystemParamItem<'w, 's, H>: CollisionHooks" in broad   # (the bound the reference's own generic systems carry)
    assert "pub fn new(collider1: Entity, collider2: Entity, contact_id: ContactId) -> Self" in contact_types and "ContactPair::new(" in hk
    assert "pub fn new(anchor1: Vector, anchor2: Vector, world_point: Vector, penetration: Scalar) -> Self" in contact_types and "ContactPoint::new(" in hk
    assert "pub fn new(points: impl IntoIterator<Item = ContactPoint>, normal: Vector) -> Self" in contact_types and "ContactManifold::new(" in hk
    for field in ("body1", "body2", "manifolds", "flags"):
        assert re.search(rf"pub {field}:", contact_types) and f"pair.{field}" in hk, field
    for field in ("friction", "restitution", "tangent_velocity", "normal", "points"):
        assert re.search(rf"pub {field}:", contact_types) and (f"manifold.{field}" in hk or f"m.{field}" in hk), field
    for field in ("anchor1", "anchor2", "penetration", "normal_speed", "feature_id1", "feature_id2"):
        assert re.search(rf"pub {field}:", contact_types) and f"p.{field}" in hk, field
    for flag in ("TOUCHING", "GENERATE_CONSTRAINTS", "STATIC1", "STATIC2", "MODIFY_CONTACTS"):
        assert f"const {flag} =" in contact_types and f"ContactPairFlags::{flag}" in hk, flag
