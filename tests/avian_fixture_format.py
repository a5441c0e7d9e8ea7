"""The `.avf` chunk stream integration/rust/avian_fixtures writes (see its src/main.rs header), its conversion to the `.npz` the tests read, a writer of the SAME
stream from a world behind the C ABI (the self-check: fixtures made from the oracle must be reproduced by the oracle and by the HIP path), and the replay that
holds a library to a fixture.  Test infrastructure."""
import struct

import numpy as np

from avian_amd import _ffi as F

BODY_W, INIT_W, JOINT_W, IMPL_W = 15, 17, 15, 13
TAGS = (b"META", b"INIT", b"JNTS", b"BODY", b"ISLD", b"PAIR", b"LIVE", b"COLR", b"IMPL")


def read_avf(path):
    """-> dict of arrays (what np.savez stores): per-frame arrays stacked, ragged ones as (flat, offsets)"""
    data = open(path, "rb").read()
    at, frames, meta, init, jnts = 0, {}, None, None, None
    while at < len(data):
        tag, frame, step, nbytes = data[at:at + 4], *struct.unpack_from("<IIQ", data, at + 4)
        payload = np.frombuffer(data, "<u4", nbytes // 4, at + 20).copy()
        at += 20 + nbytes
        assert tag in TAGS, tag
        if tag == b"META": meta = payload
        elif tag == b"INIT": init = payload.reshape(-1, INIT_W)
        elif tag == b"JNTS": jnts = payload.reshape(-1, JOINT_W)
        else: frames.setdefault(frame, {"step": step})[tag.decode()] = payload
    n = int(meta[2])
    order = sorted(frames)
    out = {"meta": meta, "init": init, "joints": jnts if jnts is not None else np.zeros((0, JOINT_W), np.uint32),
           "step": np.array([frames[f]["step"] for f in order], np.uint32),
           "body": np.stack([frames[f]["BODY"].reshape(n, BODY_W) for f in order]), "island": np.stack([frames[f]["ISLD"] for f in order])}
    for key, tag in (("pair", "PAIR"), ("live", "LIVE"), ("colr", "COLR"), ("impl", "IMPL")):
        parts = [frames[f][tag] for f in order]
        out[key] = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
        out[key + "_off"] = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    return out


def avf_to_npz(src, dst):
    np.savez_compressed(dst, **read_avf(src))


def f32(bits):
    return np.ascontiguousarray(bits, np.uint32).view(np.float32)


class Fixture:
    def __init__(self, arrays):
        self.a = arrays
        m = arrays["meta"]
        self.substeps, self.n, self.n_joints = int(m[0]), int(m[2]), int(m[3])
        self.gravity = f32(m[4:7]).astype(np.float64)
        self.step = arrays["step"]
        # the frame whose state is the world's initial condition: the last one before the first physics step (benches/src/cli.rs:358: one un-timed update first)
        self.frame0 = int(np.flatnonzero(self.step == self.step[0])[-1]) if self.step[0] == 0 else None

    def ragged(self, key, frame, width):
        o = self.a[key + "_off"]
        return self.a[key][int(o[frame]):int(o[frame + 1])].reshape(-1, width)

    def colours(self, frame):
        flat = self.ragged("colr", frame, 1).reshape(-1)
        out, at = [], 0
        for _ in range(F.GRAPH_COLOR_COUNT):
            k = int(flat[at]); out.append(flat[at + 1:at + 1 + 2 * k].reshape(k, 2)[:, 0].copy()); at += 1 + 2 * k
        return out


def load(path):
    return Fixture(dict(np.load(path)))


def world_from_fixture(lib, fx, frame=None):
    """a closed-loop world in the fixture's state at `frame` (default: its initial frame), sleeping ON as in PhysicsPlugins::default()"""
    frame = fx.frame0 if frame is None else frame
    init, b = fx.a["init"], fx.a["body"][frame]
    cfg = F.default_config(32, substeps=fx.substeps, gravity=tuple(fx.gravity))
    w = F.World(lib, cfg)
    ii = f32(init[:, 6:12]).reshape(-1, 6)   # m00 m01 m02 m11 m12 m22
    w.bodies_upload(position=f32(b[:, 0:3]), rotation=f32(b[:, 3:7]), linear_velocity=f32(b[:, 7:10]), angular_velocity=f32(b[:, 10:13]), inv_mass=f32(init[:, 5]),
                    inv_inertia_local=ii, rb_type=init[:, 0].astype(np.uint8), center_of_mass=f32(init[:, 12:15]))
    n = fx.n
    w.colliders_upload(entity_index=np.arange(n, dtype=np.uint32), body=np.arange(n, dtype=np.int32), shape=init[:, 1].astype(np.uint8), half_extents=f32(init[:, 2:5]))
    j = fx.a["joints"]
    if len(j):
        comp = np.zeros((len(j), 3), np.float32); comp[:, 0] = f32(j[:, 14])
        dist = j[:, 0] == 4
        w.joints_upload(joint_type=j[:, 0].astype(np.uint8), body1=j[:, 1].astype(np.int32), body2=j[:, 2].astype(np.int32), local_anchor1=f32(j[:, 3:6]), local_anchor2=f32(j[:, 6:9]),
                        compliance=comp, axis=f32(j[:, 9:12]), limit_min=f32(j[:, 12]), limit_max=f32(j[:, 13]), limit_flags=np.where((j[:, 0] == 3) | dist, 1, 0).astype(np.uint8),
                        collision_disabled=np.ones(len(j), np.uint8))
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.collider_materials_upload(friction=f32(init[:, 15]), restitution=f32(init[:, 16]))
    w.pipeline_enable(); w.sleeping_enable()
    return w


# ---- the same stream from a world behind the C ABI (self-check) --------------------------------------------------------------------------------------
class AvfWriter:
    def __init__(self, path):
        self.f = open(path, "wb")

    def chunk(self, tag, frame, step, payload):
        p = np.ascontiguousarray(payload, np.uint32).reshape(-1)
        self.f.write(tag + struct.pack("<IIQ", frame, step, p.size * 4) + p.tobytes())

    def close(self):
        self.f.close()


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def write_frame(wr, w, frame, step, n, seen):
    b = w.bodies_download(); st = w.sleeping_state()
    body = np.zeros((n, BODY_W), np.uint32)
    body[:, 0:3], body[:, 3:7], body[:, 7:10], body[:, 10:13] = bits(b["position"]), bits(b["rotation"]), bits(b["linear_velocity"]), bits(b["angular_velocity"])
    body[:, 13], body[:, 14] = bits(st["sleep_timer"]), st["sleeping"]
    wr.chunk(b"BODY", frame, step, body); wr.chunk(b"ISLD", frame, step, st["island"])
    pairs, ids = (w.pairs_get(), w.pipeline_new_pair_ids()) if step else (np.zeros(0, F.PAIR_DTYPE), np.zeros(0, np.uint32))
    wr.chunk(b"PAIR", frame, step, np.stack([ids, pairs["collider1"], pairs["collider2"]], axis=1) if len(ids) else np.zeros(0, np.uint32))
    off, handles = w.pipeline_handles()
    colr = []
    for c in range(F.GRAPH_COLOR_COUNT):
        h = handles[off[c]:off[c + 1]]
        colr.append(np.concatenate([[len(h)], np.stack([h, np.zeros_like(h)], axis=1).reshape(-1)]))
    wr.chunk(b"COLR", frame, step, np.concatenate(colr).astype(np.uint32))
    impl = []
    if len(handles):
        hs = np.sort(handles); c = w.contacts_download(hs)
        for k, cid in enumerate(hs):
            for p in range(int(c["point_count"][k])):
                impl.append([cid, 0, p, c["feature_id1"][k, p], c["feature_id2"][k, p], *bits([c["normal_impulse"][k, p], c["warm_start_normal_impulse"][k, p], *c["warm_start_tangent_impulse"][k, p],
                                                                                              c["penetration"][k, p], *c["normal"][k]])])
    wr.chunk(b"LIVE", frame, step, np.zeros(0, np.uint32))   # (not reproduced from the ABI: the replay does not read it)
    wr.chunk(b"IMPL", frame, step, np.array(impl, np.uint32) if impl else np.zeros(0, np.uint32))


def write_fixture_from_library(lib, sc, path, substeps, frames, joints=None):
    """a fixture in the generator's format, made from `lib` (the oracle): one frame of the initial state, then one per step"""
    n = sc.n
    w = F.World(lib, F.default_config(32, substeps=substeps))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=sc.friction, restitution=sc.restitution)
    w.pipeline_enable(); w.sleeping_enable()
    wr = AvfWriter(path)
    wr.chunk(b"META", 0, 0, np.concatenate([[substeps], bits([1.0 / 60.0]), [n, 0], bits([0.0, -9.81, 0.0])]))
    init = np.zeros((n, INIT_W), np.uint32)
    init[:, 0], init[:, 1], init[:, 2:5], init[:, 5] = sc.rb_type, sc.shape, bits(sc.half_extents), bits(sc.inv_mass)
    init[:, 6:12] = bits(sc.inv_inertia_local); init[:, 15], init[:, 16] = bits([sc.friction])[0], bits([sc.restitution])[0]
    wr.chunk(b"INIT", 0, 0, init); wr.chunk(b"JNTS", 0, 0, np.zeros(0, np.uint32))
    write_frame(wr, w, 0, 0, n, {})
    for s in range(1, frames):
        w.step(); w.synchronize()
        write_frame(wr, w, s, s, n, {})
    wr.close(); w.close()


# ---- the replay -----------------------------------------------------------------------------------------------------------------------------------------
def replay(lib, fx, steps, body_tol, exact_steps):
    """step a world made from the fixture's initial frame next to the fixture.  Integer structures (new-pair sequence with ids, colour lists with order, island ids)
    must be EQUAL and bodies within body_tol(step) for the first `exact_steps` steps; afterwards the comparison goes on until the trajectories part (a pile is chaotic:
    DESIGN.md section 2, N1) and the report says how far it held.  -> dict(report)"""
    w = world_from_fixture(lib, fx)
    f0 = fx.frame0
    held = 0
    worst = 0.0
    for k in range(1, steps + 1):
        f = f0 + k
        if f >= len(fx.step) or fx.step[f] != fx.step[f0] + k:
            break
        w.step(); w.synchronize()
        b = w.bodies_download(); fb = fx.a["body"][f]
        d = max(float(np.abs(b["position"] - f32(fb[:, 0:3])).max()), float(np.abs(b["linear_velocity"] - f32(fb[:, 7:10])).max()),
                float(np.abs(b["angular_velocity"] - f32(fb[:, 10:13])).max()), float(np.abs(np.abs(b["rotation"]) - np.abs(f32(fb[:, 3:7]))).max()))
        worst = max(worst, d)
        pairs, ids = w.pairs_get(), w.pipeline_new_pair_ids()
        mine = np.stack([ids, pairs["collider1"], pairs["collider2"]], axis=1).astype(np.uint32) if len(ids) else np.zeros((0, 3), np.uint32)
        same_pairs = np.array_equal(mine, fx.ragged("pair", f, 3))
        off, handles = w.pipeline_handles()
        same_colours = all(np.array_equal(handles[off[c]:off[c + 1]], want) for c, want in enumerate(fx.colours(f)))
        same_islands = np.array_equal(w.sleeping_state()["island"], fx.a["island"][f])
        ok = same_pairs and same_colours and same_islands and d <= body_tol(k)
        if k <= exact_steps:
            assert same_pairs, f"step {k}: the broad phase's new pairs (ContactId, collider1, collider2 in emission order) differ from the fixture"
            assert same_colours, f"step {k}: a GraphColor's manifold_handles differ from the fixture (content or order)"
            assert same_islands, f"step {k}: island ids differ from the fixture"
            assert d <= body_tol(k), f"step {k}: bodies differ from the fixture by {d:.3g} (tolerance {body_tol(k):.3g})"
        if not ok:
            break
        held = k
    w.close()
    return {"steps_held": held, "worst_body_difference": worst}
