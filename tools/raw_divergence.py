"""Where does a free run first leave the reference in BULLET units?  The exchange format is in uu: two Bullet-unit values one ulp apart can
round to the same uu float, so a tape that is bit-identical in the exchanged fields can already differ underneath.  Runs the LIVE reference
(oracle/_ref/libref_oracle.so, needs /root/reference's build) and the host build of the stepper over a tape of the two-file mesh fixture
(tests/golden/seam_golden.npz) or of the main fixture and prints, per tick, the first raw field (pos / rot / vel / angvel of ball and
cars) whose bits differ.      usage: raw_divergence.py <seam|main|gym> <scenario or gym case> [ticks]"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simlib import PortSim, RefSim, write_cmf_parts
from rlgymppo_cpp_amd.state import ArenaState

which, name = sys.argv[1], sys.argv[2]
gold = np.load(os.path.join(ROOT, "tests", "golden", {"seam": "seam_golden.npz", "tess": "tess_golden.npz"}.get(which, "sim_golden.npz")))
if which == "gym":
    # a recorded gym rollout as a control tape, up to its first episode end: Gym::Step runs ONE tick on the previous step's controls, takes
    # the GameState, then sets the new controls and runs tickSkip - 1 more (Gym.cpp:68-102); the first step's "previous" controls are zeros
    acts, dones = gold[f"gym/{name}/actions"], gold[f"gym/{name}/done"]
    n_steps = int(np.argmax(dones)) if dones.any() else len(acts)
    st0 = ArenaState.from_buffer_copy(gold[f"gym/{name}/start_raw" if f"gym/{name}/start_raw" in gold.files else f"gym/{name}/start"].tobytes()); nc = st0.num_cars
    table = np.zeros((90, 8), np.float32); _pl = PortSim().lib; _pl.port_action_table.argtypes = [C.c_void_p]; _pl.port_action_table(table.ctypes.data)
    skip = int(gold[f"gym/{name}/cfg"][1])
    tape = np.zeros((n_steps * skip, nc, 8), np.float32)
    for s_ in range(n_steps):
        for k in range(nc):
            tape[s_ * skip + 1:(s_ + 1) * skip, k] = table[acts[s_, k]]
            if s_ + 1 < n_steps:
                tape[(s_ + 1) * skip, k] = table[acts[s_, k]]
    print(f"gym rollout {name}: {n_steps} steps x {skip} ticks before its first episode end")
else:
    tape = np.ascontiguousarray(gold[f"phys/{name}/tape"], np.float32)
    st0 = ArenaState.from_buffer_copy(gold[f"phys/{name}/start_raw"].tobytes()); nc = st0.num_cars
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else len(tape)
verts, tris = gold["mesh_verts"], gold["mesh_tris"]
port = PortSim()
if which in ("seam", "tess"):
    parts = gold["mesh_parts"]; port.set_mesh(verts, tris, parts)
    root = tempfile.mkdtemp(prefix="rawdiv_"); write_cmf_parts(verts, tris, parts, root)
    ref = RefSim(verts, tris, mesh_dir=root)
else:
    port.set_mesh(verts, tris); ref = RefSim(verts, tris)
# the live arena visits its cars in ITS unordered_set's order (heap addresses of this process), not in the recording arena's: the port follows it
# RAWDIV_REHASH=<buckets> [RAWDIV_SHUFFLE=<seed>]: another car order in the live arena -- Arena::_cars is an unordered_set of pointers, so
# its iteration order changes with the bucket count (ref_arena_rehash) and with where the cars lie on the heap (ref_arena_new_shuffled);
# e.g. RAWDIV_REHASH=1 gives 654321 where the plain arena has 123456
_seed = int(os.environ.get("RAWDIV_SHUFFLE", "0")); _buckets = int(os.environ.get("RAWDIV_REHASH", "0"))
ref.lib.ref_arena_new_shuffled.restype = C.c_void_p; ref.lib.ref_arena_new_shuffled.argtypes = [C.c_int, C.c_uint]
ref.lib.ref_arena_rehash.argtypes = [C.c_void_p, C.c_int]
def _new_arena(n):
    h = C.c_void_p(ref.lib.ref_arena_new_shuffled(n, _seed))
    if _buckets: ref.lib.ref_arena_rehash(h, _buckets)
    return h
a = _new_arena(nc // 2); ref.set_state(a, st0)
st0.car_order = ref.get_state(a).car_order
print("car order of the live reference arena: %x" % st0.car_order)
# port: raw state after every tick
raw_p = np.zeros((ticks, 1 + nc, 18), np.float32)
st = ArenaState.from_buffer_copy(bytes(st0))
port.lib.port_run_tape_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
port.lib.port_run_tape_raw(C.byref(st), tape.ctypes.data, ticks, raw_p.ctypes.data)
# reference: the same tape, tick by tick
ref.lib.ref_arena_get_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
raw_r = np.zeros((ticks, 1 + nc, 18), np.float32)
for t in range(ticks):
    for k in range(nc):
        ref.set_controls(a, k, tape[t, k])
    ref.step(a, 1)
    ref.lib.ref_arena_get_raw(a, nc, raw_r[t].ctypes.data)
names = ["pos.x", "pos.y", "pos.z"] + ["rot[%d][%d]" % (r, c) for r in range(3) for c in range(3)] + ["vel.x", "vel.y", "vel.z", "angvel.x", "angvel.y", "angvel.z"]
bp, br = raw_p.view(np.uint32), raw_r.view(np.uint32)
first = None
for t in range(ticks):
    d = np.argwhere(bp[t] != br[t])
    if len(d):
        if first is None:
            first = t
            print(f"first raw difference after tick {t + 1}:")
        if t < first + 3:
            for b, f in d[:20]:
                print(f"   tick {t + 1} body {'ball' if b == 0 else 'car%d' % (b - 1)} {names[f]}: port {raw_p[t, b, f]!r} ({bp[t, b, f]:08x}) reference {raw_r[t, b, f]!r} ({br[t, b, f]:08x})")
if first is None:
    print(f"{name}: raw Bullet-unit state bit-identical for all {ticks} ticks")

if first is not None and os.environ.get("RAWDIV_CONTACTS", "1") == "1":
    # the contacts of the first differing tick on both sides (the reference's: world points on A / B; the port's: points relative to the bodies)
    T = first + 1
    st = ArenaState.from_buffer_copy(bytes(st0))
    buf = np.zeros((64, 16), np.float32)
    buf2 = np.zeros((64, 16), np.float32)
    wp = np.zeros((nc, 4, 12), np.float32)
    port.lib.port_run_tape_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; port.lib.port_run_tape_contacts.restype = C.c_int
    n = port.lib.port_run_tape_contacts(C.byref(st), tape.ctypes.data, T, buf.ctypes.data, 64, buf2.ctypes.data, wp.ctypes.data)
    print(f"port contacts of tick {T} (a, b, sid | ra | rb | normal | dist | applied):")
    for r in buf[:n]:
        print("   a %2d b %2d sid %d | ra (%.6f %.6f %.6f) | rb (%.6f %.6f %.6f) | n (%.7f %.7f %.7f) | dist %.7g | applied %.7g" % (r[0], r[1], r[2], *r[4:7], *r[7:10], *r[10:13], r[13], r[14]))
    for q in buf2[:n]:
        print("      friction dir (%.7f %.7f %.7f) applied %.7g | normal rhs %.7g jac %.7g | ext_f (%.7g %.7g %.7g) ext_t (%.7g %.7g %.7g) | friction rhs %.7g jac %.7g" % (*q[0:3], q[3], q[4], q[5], *q[6:9], *q[9:12], q[12], q[13]))
    a2 = _new_arena(nc // 2); ref.set_state(a2, st0)
    if ref.get_state(a2).car_order != st0.car_order:
        print("   (the second reference arena visits its cars in another order, %x: its dumps below may not belong to the same run)" % ref.get_state(a2).car_order)
    for t in range(T):
        for k in range(nc):
            ref.set_controls(a2, k, tape[t, k])
        ref.step(a2, 1)
        if t == T - 2:
            pre = np.zeros((1 + nc, 18), np.float32); ref.lib.ref_arena_get_raw(a2, nc, pre.ctypes.data)
    rb = np.zeros((64, 16), np.float32)
    ref.lib.ref_debug_manifolds.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; ref.lib.ref_debug_manifolds.restype = C.c_int
    n = ref.lib.ref_debug_manifolds(a2, rb.ctypes.data, 64)
    print(f"reference manifold points after tick {T} (body0, body1, manifold | world point on A | on B | normal on B | dist | applied):")
    for r in rb[:n]:
        print("   b0 %2d b1 %2d m %d | A (%.6f %.6f %.6f) | B (%.6f %.6f %.6f) | n (%.7f %.7f %.7f) | dist %.7g | applied %.7g" % (r[0], r[1], r[2], *r[4:7], *r[7:10], *r[10:13], r[13], r[14]))
    fb = np.zeros((64, 8), np.float32)
    ref.lib.ref_debug_manifold_friction.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; ref.lib.ref_debug_manifold_friction.restype = C.c_int
    nf = ref.lib.ref_debug_manifold_friction(a2, fb.ctypes.data, 64)
    for q in fb[:nf]:
        print("      friction dir (%.7f %.7f %.7f) applied %.7g | mu %.4g restitution %.4g | applied normal %.7g" % (*q[0:3], q[3], q[4], q[5], q[6]))
    # wheels of that tick: suspension length, relative velocity, contact point / normal, |friction impulse| (the port's column 1 = what the ray hit: -1 none, 0 world, 1 ball, 2 + i car i)
    ref.lib.ref_debug_wheels.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for k in range(nc):
        wr = np.zeros((4, 12), np.float32); ref.lib.ref_debug_wheels(a2, k, wr.ctypes.data)
        cols = [0, 2, 3, 4, 5, 6, 7, 8, 9, 11]
        if not np.array_equal(wp[k][:, cols], wr[:, cols]):   # (values: a -0 against a +0 is not a difference here)
            for w in range(4):
                print("   car%d wheel %d port: susp %.7g relvel %.7g inv %.7g cp (%.6f %.6f %.6f) n (%.6f %.6f %.6f) |imp| %.7g hit %d" % (k, w, wp[k, w, 0], wp[k, w, 2], wp[k, w, 3], *wp[k, w, 4:7], *wp[k, w, 7:10], wp[k, w, 11], int(wp[k, w, 1])))
                print("   car%d wheel %d ref : susp %.7g relvel %.7g inv %.7g cp (%.6f %.6f %.6f) n (%.6f %.6f %.6f) |imp| %.7g" % (k, w, wr[w, 0], wr[w, 2], wr[w, 3], *wr[w, 4:7], *wr[w, 7:10], wr[w, 11]))
    print("car0 before that tick (reference): origin", pre[1, :3], "vel", pre[1, 12:15], "angvel", pre[1, 15:18])

if first is not None:
    # the EXCHANGED state (uu, flags, contact normals, timers ...) one tick before the first raw difference: anything the raw dump does not cover
    T0 = first   # ticks run
    st = ArenaState.from_buffer_copy(bytes(st0))
    port.lib.port_run_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    port.lib.port_run_tape(C.byref(st), tape.ctypes.data, T0, 1 << 30, None)
    a3 = _new_arena(nc // 2); ref.set_state(a3, st0)
    for t in range(T0):
        for k in range(nc):
            ref.set_controls(a3, k, tape[t, k])
        ref.step(a3, 1)
    sr = ref.get_state(a3)
    for k in range(nc):
        cp, cr = st.cars[k], sr.cars[k]
        for fld, _ in cp._fields_:
            vp, vr = getattr(cp, fld), getattr(cr, fld)
            bp_, br_ = bytes(vp) if hasattr(vp, "_length_") else bytes(C.c_double(vp)) if isinstance(vp, float) else bytes(C.c_int64(int(vp))), None
            br_ = bytes(vr) if hasattr(vr, "_length_") else bytes(C.c_double(vr)) if isinstance(vr, float) else bytes(C.c_int64(int(vr)))
            if bp_ != br_:
                print(f"   exchanged state after tick {T0}: car{k}.{fld}: port {list(vp) if hasattr(vp, '_length_') else vp} reference {list(vr) if hasattr(vr, '_length_') else vr}")
    print(f"   (exchanged car fields compared after tick {T0})")
