// oracle/arena_port.cpp — TEST INFRASTRUCTURE ONLY.
//
// Host (g++) build of the arena stepper's scalar core.  The physics tick is ~3k lines of branchy fp32 code
// that has to be developed and debugged in a container without a GPU, so the tick is written ONCE in
// rlgymppo_cpp_amd/csrc/arena_*.h as host/device-neutral inline functions; this file compiles those headers
// for the CPU and exposes them to the tests.  It is the "port" CPU restatement of the stepper:
//   * pinned against the REAL reference (oracle/_ref, golden trajectories under tests/golden/) by
//     tests/test_oracle_golden.py, and
//   * used by the -m gpu tests as the tick-by-tick oracle of the HIP kernel (same source, different
//     compiler/ISA: agreement is to fp32 rounding of libm calls).
// The dependency runs oracle -> product headers only; nothing in rlgymppo_cpp_amd/ includes, links or calls
// anything in oracle/ (the product path fails loudly without its HIP library).
// EPA usage statistics of the host build (tools/gjk_fuzz.py, tools/epa_stats.py): runs, max / histogram of support vertices and faces, statuses
static int g_epa_stats[64];   // [0] runs, [1] max verts, [2] max faces, [3] max iterations, [4 + status] status counts, [16 + min(verts / 4, 31)] histogram
#define RLG_EPA_STATS(nv, nf, it, st) do { g_epa_stats[0]++; if ((nv) > g_epa_stats[1]) g_epa_stats[1] = (nv); if ((nf) > g_epa_stats[2]) g_epa_stats[2] = (nf); \
    if ((it) > g_epa_stats[3]) g_epa_stats[3] = (it); g_epa_stats[4 + ((st) < 10 ? (st) : 10)]++; g_epa_stats[16 + ((nv) / 4 < 31 ? (nv) / 4 : 31)]++; } while (0)
static long g_port_counts[16];   // events of the stepper's bookkeeping (RLG_DBG_COUNT; 8 = env-ticks redone with the big contact layout, 7 = contacts lost: never)
#define RLG_DBG_COUNT(i) (void)(__atomic_fetch_add(&g_port_counts[(i) & 15], 1L, __ATOMIC_RELAXED))
#include "../rlgymppo_cpp_amd/csrc/arena_gym.h"
#include "../rlgymppo_cpp_amd/csrc/arena_mesh.h"
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

using namespace rlg;

// The contact layout the port steps with (arena_contact.h): the big one -- a manifold for every mesh object, a slot for every pair, a row for every
// slot: nothing can overflow, which is what makes the port the oracle of the device's small layout + fallback.  -DORACLE_SMALL_LAYOUT builds
// the small layout with its fallback instead (with -DRLG_TINY_LAYOUT: one that overflows all the time; tests/test_oracle_golden.py holds both
// builds to the same fixtures).
#ifdef ORACLE_SMALL_LAYOUT
template <int NC> using PortWork = TickWork<NC, 0>;
#else
template <int NC> using PortWork = TickWork<NC, 1>;
#endif

namespace {
HostMesh g_mesh;
MeshView view() {
    MeshView v; v.nodes = g_mesh.nodes.data(); v.tris = g_mesh.tris.data(); v.nodes_fast = nullptr;
    v.n_nodes = (int)g_mesh.nodes.size(); v.n_tris = (int)g_mesh.tris.size(); v.n_fast = 0;
    v.grid = g_mesh.grid.empty() ? nullptr : g_mesh.grid.data();
    v.bp = g_mesh.grid.size() > (size_t)GRID_WORDS ? g_mesh.grid.data() + GRID_WORDS : nullptr;
    return v;
}
template <int NC>
void step_t(RlgpuArenaState* s, int ticks, uint32_t seed, uint32_t env) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    for (int t = 0; t < ticks; t++) { TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, seed, env, ev, W); }
    arena_to_host(A, G, *s);
}
// ... the same with the history in the caller's hands: hist[8] (all zero = a fresh arena) goes in before the ticks and comes back after them
template <int NC>
void step_hist_t(RlgpuArenaState* s, int ticks, uint16_t* hist) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    for (int b = 0; b <= NC; b++) A.bp_hist[b] = hist[b];
    MeshView mv = view();
    PortWork<NC> W;
    for (int t = 0; t < ticks; t++) { TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, 0, 0, ev, W); }
    for (int b = 0; b <= NC; b++) hist[b] = A.bp_hist[b];
    arena_to_host(A, G, *s);
}
// n states stepped one tick each, one after the other, in ONE arena: what the state does not carry -- the broadphase's memory of where its
// proxies were and in which order they arrived (Arena::bp_hist) -- passes from each to the next, as in an arena of the reference that is
// set_state'd and stepped again and again (how the one-tick fixtures were recorded: tests/golden/make_sim_golden.py, the pair arena)
template <int NC>
void step_chain_t(RlgpuArenaState* s, int n) {
    Arena<NC> A; GymEnv<NC> G;
    MeshView mv = view();
    PortWork<NC> W;
    uint16_t hist[NC + 1] = {};
    for (int i = 0; i < n; i++) {
        arena_from_host(A, G, s[i]);
        for (int b = 0; b <= NC; b++) A.bp_hist[b] = hist[b];
        TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, 0, 0, ev, W);
        for (int b = 0; b <= NC; b++) hist[b] = A.bp_hist[b];
        arena_to_host(A, G, s[i]);
    }
}
// a whole control tape without leaving the stepper's own units: the state is converted once, every tick takes its controls from the tape
// ([ticks][nc][8], the reference's CarControls order), and every `every`-th tick a copy is written out.  (port_arena_step tick by tick
// rounds the state to uu and back between ticks, which the reference's free-running arena does not do.)
template <int NC>
void run_tape_t(RlgpuArenaState* s, const float* tape, int ticks, int every, RlgpuArenaState* out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    int n_out = 0;
    for (int t = 0; t < ticks; t++) {
        for (int k = 0; k < NC; k++) A.cars[k].ctl = ctl_from(tape + ((size_t)t * NC + k) * 8);
        TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, 0, 0, ev, W);
        if (out && (t + 1) % every == 0) { out[n_out] = *s; arena_to_host(A, G, out[n_out]); n_out++; }
    }
    arena_to_host(A, G, *s);
}
// the same run, writing every tick's Bullet-unit state: per body 18 floats = pos[3], rot rows[9], vel[3], angvel[3]; raw_out: ticks x (1 + NC) x 18
template <int NC>
void run_tape_raw_t(RlgpuArenaState* s, const float* tape, int ticks, float* raw_out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    auto put = [](float* o, const Body& b) {
        o[0] = b.pos.x; o[1] = b.pos.y; o[2] = b.pos.z;
        const V3 r0 = b.rot.r0, r1 = b.rot.r1, r2 = b.rot.r2;
        o[3] = r0.x; o[4] = r0.y; o[5] = r0.z; o[6] = r1.x; o[7] = r1.y; o[8] = r1.z; o[9] = r2.x; o[10] = r2.y; o[11] = r2.z;
        o[12] = b.vel.x; o[13] = b.vel.y; o[14] = b.vel.z; o[15] = b.angvel.x; o[16] = b.angvel.y; o[17] = b.angvel.z;
    };
    for (int t = 0; t < ticks; t++) {
        for (int k = 0; k < NC; k++) A.cars[k].ctl = ctl_from(tape + ((size_t)t * NC + k) * 8);
        TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, 0, 0, ev, W);
        float* o = raw_out + (size_t)t * (1 + NC) * 18;
        put(o, A.ball.b);
        for (int k = 0; k < NC; k++) {
            Body b = A.cars[k].b;
            if (A.cars[k].flags & CF_IS_DEMOED) b.rot = car_ghost_rot(A.cars[k]);     // the rigid body's own basis (the reference's raw dump reads the body)
            put(o + 18 * (1 + k), b);
        }
    }
    arena_to_host(A, G, *s);
}
// ... and the contact list of the LAST tick as the solver saw it (the 16 floats of debug_tick_t), for a tick reached inside the stepper's own units
template <int NC>
int run_tape_contacts_t(RlgpuArenaState* s, const float* tape, int ticks, float* out, int cap, float* out2, float* wheels_out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    for (int t = 0; t < ticks; t++) {
        for (int k = 0; k < NC; k++) A.cars[k].ctl = ctl_from(tape + ((size_t)t * NC + k) * 8);
        TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, 0, 0, ev, W);
    }
    int n = 0;
    for (int k = 0; k < W.L.n && n < cap; k++) {
        const Contact& c = W.L.c[W.cidx[k]];
        float* o = out + 16 * n++;
        o[0] = (float)c.a; o[1] = (float)c.b; o[2] = (float)c.sid; o[3] = c.special ? 1.f : 0.f;
        o[4] = c.ra.x; o[5] = c.ra.y; o[6] = c.ra.z; o[7] = c.rb.x; o[8] = c.rb.y; o[9] = c.rb.z;
        o[10] = c.n.x; o[11] = c.n.y; o[12] = c.n.z; o[13] = c.dist;
        o[14] = W.nrow[k] >= 0 ? W.R[W.nrow[k]].applied : 0.f; o[15] = contact_friction(c, A.mut);
        if (out2) {   // the rows: friction direction[3], applied friction impulse, normal rhs, normal jac, body a's external impulses ext_f[3] / ext_t[3], friction rhs, friction jac, v.x, w.y
            float* q = out2 + 16 * (n - 1);
            for (int i = 0; i < 16; i++) q[i] = 0.f;
            const int fr = W.frow[k], nr = W.nrow[k];
            if (fr >= 0) { q[0] = W.R[fr].n1.x; q[1] = W.R[fr].n1.y; q[2] = W.R[fr].n1.z; q[3] = W.R[fr].applied; }
            if (nr >= 0) { q[4] = W.R[nr].rhs; q[5] = W.R[nr].jac; }
            const SolverBody& sb = W.B[c.a];
            q[6] = sb.ext_f.x; q[7] = sb.ext_f.y; q[8] = sb.ext_f.z; q[9] = sb.ext_t.x; q[10] = sb.ext_t.y; q[11] = sb.ext_t.z;
            if (fr >= 0) { q[12] = W.R[fr].rhs; q[13] = W.R[fr].jac; }
            q[14] = sb.v.x; q[15] = sb.w.y;
        }
    }
    if (wheels_out)   // every car's wheel scratch of that last tick, laid out like ref_debug_wheels (4 x 12 floats per car; [1] is unused)
        for (int k = 0; k < NC; k++) for (int w = 0; w < 4; w++) {
            const WheelTmp& wt = W.ctx[k].w[w];
            float* o = wheels_out + (size_t)(k * 4 + w) * 12;
            o[0] = wt.susp_len; o[1] = (float)wt.ground; o[2] = wt.susp_rel_vel; o[3] = wt.clipped_inv;
            o[4] = wt.contact_point.x; o[5] = wt.contact_point.y; o[6] = wt.contact_point.z;
            o[7] = wt.contact_normal.x; o[8] = wt.contact_normal.y; o[9] = wt.contact_normal.z;
            o[10] = wt.hard_point.z; o[11] = len(wt.impulse);
        }
    arena_to_host(A, G, *s);
    return n;
}
}  // namespace

extern "C" {

int port_run_tape_contacts(RlgpuArenaState* s, const float* tape, int ticks, float* out, int cap, float* out2, float* wheels_out) {
    if (s->num_cars == 2) return run_tape_contacts_t<2>(s, tape, ticks, out, cap, out2, wheels_out); else if (s->num_cars == 4) return run_tape_contacts_t<4>(s, tape, ticks, out, cap, out2, wheels_out); else return run_tape_contacts_t<6>(s, tape, ticks, out, cap, out2, wheels_out);
}
void port_step_hist(RlgpuArenaState* s, int ticks, uint16_t* hist8) {
    if (s->num_cars == 2) step_hist_t<2>(s, ticks, hist8); else if (s->num_cars == 4) step_hist_t<4>(s, ticks, hist8); else step_hist_t<6>(s, ticks, hist8);
}
void port_step_chain(RlgpuArenaState* s, int n) {
    if (n <= 0) return;
    if (s->num_cars == 2) step_chain_t<2>(s, n); else if (s->num_cars == 4) step_chain_t<4>(s, n); else step_chain_t<6>(s, n);
}
void port_run_tape_raw(RlgpuArenaState* s, const float* tape, int ticks, float* raw_out) {
    if (s->num_cars == 2) run_tape_raw_t<2>(s, tape, ticks, raw_out); else if (s->num_cars == 4) run_tape_raw_t<4>(s, tape, ticks, raw_out); else run_tape_raw_t<6>(s, tape, ticks, raw_out);
}
void port_run_tape(RlgpuArenaState* s, const float* tape, int ticks, int every, RlgpuArenaState* out) {
    if (s->num_cars == 2) run_tape_t<2>(s, tape, ticks, every, out); else if (s->num_cars == 4) run_tape_t<4>(s, tape, ticks, every, out); else run_tape_t<6>(s, tape, ticks, every, out);
}

int port_state_size() { return (int)sizeof(RlgpuArenaState); }
// RLG_DBG_COUNT events since the last reset (16 slots)
void port_debug_counts(long* out16, int reset) { memcpy(out16, g_port_counts, sizeof(g_port_counts)); if (reset) memset(g_port_counts, 0, sizeof(g_port_counts)); }
void port_epa_stats(int* out64, int reset) { memcpy(out64, g_epa_stats, sizeof(g_epa_stats)); if (reset) memset(g_epa_stats, 0, sizeof(g_epa_stats)); }

void port_set_mesh(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris) {
    g_mesh = build_host_mesh(verts_uu, n_verts, tris, n_tris);
}
// a mesh of several objects (.cmf files): part_tris[k] triangles each, in input order (what rlgpu_env_load_cmf_dir hands to the same builder)
void port_set_mesh_parts(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris, const int32_t* part_tris, int n_parts) {
    std::vector<int> parts(part_tris, part_tris + n_parts);
    g_mesh = build_host_mesh(verts_uu, n_verts, tris, n_tris, &parts);
}
int port_procedural_mesh(float* verts_out, int cap_verts, int32_t* tris_out, int cap_tris, int* n_verts, int* n_tris) {
    std::vector<float> v; std::vector<int32_t> t;
    make_procedural_soccar(v, t);
    *n_verts = (int)v.size() / 3; *n_tris = (int)t.size() / 3;
    if (*n_verts > cap_verts || *n_tris > cap_tris) return -1;
    memcpy(verts_out, v.data(), v.size() * 4); memcpy(tris_out, t.data(), t.size() * 4);
    return 0;
}
// the host build of csrc/arena_gjk.h:gjk_box_triangle for the Octane hitbox (core, margin as the tick uses them) against one triangle
// with margin 0: out8 = normal[3], point on the triangle side[3], distance, deep flag.  Returns 1 when a point was reported.
int port_gjk_box_triangle(const float* pos3, const float* rot9, const float* tri9, float breaking, float* out8) {
    MeshTri t; memset(&t, 0, sizeof(t));
    float* p = &t.v0x; for (int k = 0; k < 9; k++) p[k] = tri9[k];
    t.edge_angle[0] = t.edge_angle[1] = t.edge_angle[2] = 6.283185307179586232f;
    M3 R = m3_rows(v3(rot9[0], rot9[1], rot9[2]), v3(rot9[3], rot9[4], rot9[5]), v3(rot9[6], rot9[7], rot9[8]));
    GjkOut g; g.n = v3(0, 0, 0); g.pb = v3(0, 0, 0); g.dist = 0.f; bool deep = false;
    const bool hit = gjk_box_triangle(v3(pos3[0], pos3[1], pos3[2]), R, hitbox_core(), BOX_MARGIN, t, breaking, g, deep);
    out8[0] = g.n.x; out8[1] = g.n.y; out8[2] = g.n.z; out8[3] = g.pb.x; out8[4] = g.pb.y; out8[5] = g.pb.z; out8[6] = g.dist; out8[7] = deep ? 1.f : 0.f;
    return hit ? 1 : 0;
}
// the half extents btBoxShape's constructor is given for the hitbox (it keeps them minus 0.04 as the core: btBoxShape.h:82-90)
void port_hitbox_ctor_half(float* out3) { const V3 h = hitbox_core(); out3[0] = h.x + 0.04f; out3[1] = h.y + 0.04f; out3[2] = h.z + 0.04f; }
// the host build of csrc/arena_world.h:box_box_ode for two Octane hitboxes (box 1 = the manifold's body0): out = n x (normal[3], point[3], dist)
int port_box_box(const float* pos1, const float* rot1, const float* pos2, const float* rot2, float* out) {
    M3 R1 = m3_rows(v3(rot1[0], rot1[1], rot1[2]), v3(rot1[3], rot1[4], rot1[5]), v3(rot1[6], rot1[7], rot1[8]));
    M3 R2 = m3_rows(v3(rot2[0], rot2[1], rot2[2]), v3(rot2[3], rot2[4], rot2[5]), v3(rot2[6], rot2[7], rot2[8]));
    Cand cs[4]; int nc = 0;
    box_box_ode(v3(pos1[0], pos1[1], pos1[2]), R1, v3(pos2[0], pos2[1], pos2[2]), R2, hitbox_half(), cs, nc);
    for (int k = 0; k < nc; k++) { float* o = out + 7 * k; o[0] = cs[k].n.x; o[1] = cs[k].n.y; o[2] = cs[k].n.z; o[3] = cs[k].pb.x; o[4] = cs[k].pb.y; o[5] = cs[k].pb.z; o[6] = cs[k].dist; }
    return nc;
}
// the host build of csrc/arena_simplex.h:ray_convex_cast as the wheel rays use it (arena_world.h:ray_ball_and_cars): out4 = fraction, normal as
// btCollisionWorld::rayTestSingleInternal reports it (normalised once more); returns 1 on a hit
int port_ray_convex(const float* from3, const float* to3, const float* half3, float radius, const float* pos3, const float* rot9, float* out4) {
    M3 R = m3_rows(v3(rot9[0], rot9[1], rot9[2]), v3(rot9[3], rot9[4], rot9[5]), v3(rot9[6], rot9[7], rot9[8]));
    RayHit best; best.kind = -1; best.frac = 1.f; best.normal = v3(0, 0, 0);
    // half3 is what btBoxShape's constructor is given: the shape keeps (half - 0.04) and a margin of 0.1 * the smallest half extent (setSafeMargin),
    // and its localGetSupportingVertex adds the two (btBoxShape.cpp:19-35, btBoxShape.h:47-56) -- arena_body.h:hitbox_core / BOX_MARGIN / hitbox_half
    const float mg = 0.1f * fminf(half3[0], fminf(half3[1], half3[2]));
    const V3 half = v3((half3[0] - 0.04f) + mg, (half3[1] - 0.04f) + mg, (half3[2] - 0.04f) + mg);
    const bool hit = ray_convex_hit(v3(from3[0], from3[1], from3[2]), v3(to3[0], to3[1], to3[2]), R, v3(pos3[0], pos3[1], pos3[2]), half, radius, 7, best);
    out4[0] = best.frac; out4[1] = best.normal.x; out4[2] = best.normal.y; out4[3] = best.normal.z;
    return hit ? 1 : 0;
}
// the static world as a wheel ray meets it (csrc/arena_world.h: ray_planes, then the mesh with Bullet's leaf admission for edge-tolerance hits): out4 = fraction,
// normal; returns 1 on a hit.  from / to in Bullet units.  (tests/test_live_reference.py: rays beside triangle edges against btCollisionWorld::rayTest)
int port_ray_world(const float* from3, const float* to3, float* out4) {
    const V3 from = v3(from3[0], from3[1], from3[2]), to = v3(to3[0], to3[1], to3[2]);
    RayHit best = ray_planes(from, to);
    ray_mesh_walk(view(), from, to, best);
    out4[0] = best.frac; out4[1] = best.normal.x; out4[2] = best.normal.y; out4[3] = best.normal.z;
    return best.kind >= 0 ? 1 : 0;
}
// debugging probe: every stored triangle whose plane the segment crosses -- out per triangle: stored index, source index, hit fraction with / without the leaf admission (2 = no hit), admitted flag
int port_ray_debug(const float* from3, const float* to3, float* out, int cap) {
    const V3 from = v3(from3[0], from3[1], from3[2]), to = v3(to3[0], to3[1], to3[2]);
    MeshView mv = view(); int n = 0;
    for (int i = 0; i < mv.n_tris && n < cap; i++) {
        const MeshTri& t = mv.tris[i];
        const V3 v0 = v3(t.v0x, t.v0y, t.v0z), v1 = v3(t.v1x, t.v1y, t.v1z), v2 = v3(t.v2x, t.v2y, t.v2z);
        RayHit a; a.kind = -1; a.frac = 1.f; a.normal = v3(0, 0, 0); RayHit b = a;
        ray_triangle(v0, v1, v2, from, to, a, mv.bp, &t);
        ray_triangle(v0, v1, v2, from, to, b);
        if (b.kind != 0) continue;
        float* o = out + 5 * n++;
        o[0] = (float)i; o[1] = (float)g_mesh.source_tri[i]; o[2] = a.kind == 0 ? a.frac : 2.f; o[3] = b.frac; o[4] = ray_leaf_admits(mesh_leaf_frame(mv.bp, t.obj), v0, v1, v2, from, to) ? 1.f : 0.f;
    }
    return n;
}
// the host build of csrc/arena_world.h:adjust_internal_edge on stored triangle `stored_index` of the mesh port_set_mesh built
// (g_mesh.source_tri maps it to the input's numbering): out7 = normal[3], point on the triangle[3], distance
int port_adjust_internal_edge(int stored_index, const float* pb3, const float* n3, float dist, float* out7) {
    if (stored_index < 0 || stored_index >= (int)g_mesh.tris.size()) return -1;
    V3 pb = v3(pb3[0], pb3[1], pb3[2]), n = v3(n3[0], n3[1], n3[2]);
    adjust_internal_edge(g_mesh.tris[stored_index], pb, n, dist);
    out7[0] = n.x; out7[1] = n.y; out7[2] = n.z; out7[3] = pb.x; out7[4] = pb.y; out7[5] = pb.z; out7[6] = dist;
    return 0;
}
int port_mesh_triangle(int stored_index, float* tri9, uint32_t* flags, float* angles3) {
    if (stored_index < 0 || stored_index >= (int)g_mesh.tris.size()) return -1;
    const MeshTri& t = g_mesh.tris[stored_index];
    const float* p = &t.v0x; for (int k = 0; k < 9; k++) tri9[k] = p[k];
    *flags = t.edge_flags; for (int k = 0; k < 3; k++) angles3[k] = t.edge_angle[k];
    return g_mesh.source_tri[stored_index];
}
int port_mesh_visit_order(int32_t* out, int cap) { int n = (int)g_mesh.source_tri.size(); for (int i = 0; i < n && i < cap; i++) out[i] = g_mesh.source_tri[i]; return n; }
int port_mesh_counts(int* n_nodes, int* n_tris) { *n_nodes = (int)g_mesh.nodes.size(); *n_tris = (int)g_mesh.tris.size(); return 0; }

// advance the physical state by `ticks` ticks with the controls stored in the state
void port_arena_step(RlgpuArenaState* s, int ticks, uint32_t seed, uint32_t env) {
    if (s->num_cars == 2) step_t<2>(s, ticks, seed, env);
    else if (s->num_cars == 4) step_t<4>(s, ticks, seed, env);
    else if (s->num_cars == 6) step_t<6>(s, ticks, seed, env);
}

}  // extern "C"

// ---- gym layer (same GymConfig struct as the C-ABI's RlgpuGymConfig) ------------------------------------------
static float g_action_table[90 * 8];
static bool g_table_built = false;
static const float* table() { if (!g_table_built) { build_action_table(g_action_table); g_table_built = true; } return g_action_table; }

template <int NC>
static void gym_reset_t(RlgpuArenaState* s, const GymConfig* cfg, uint32_t env, float* obs, int run_setter) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    gym_reset_env<NC>(A, G, *cfg, env, obs, (size_t)obs_size<NC>(*cfg), run_setter != 0);
    arena_to_host(A, G, *s);
}
template <int NC>
static void gym_step_t(RlgpuArenaState* s, const GymConfig* cfg, uint32_t env, const int32_t* actions, float* obs, float* rew, int32_t* done, uint16_t* hist = nullptr) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    if (hist) for (int b = 0; b <= NC; b++) A.bp_hist[b] = hist[b];      // the arena's broadphase history, kept by the caller (port_step_hist)
    MeshView mv = view();
    PortWork<NC> W;
    gym_step_env<NC>(A, G, *cfg, mv, table(), actions, env, obs, (size_t)obs_size<NC>(*cfg), rew, done, W);
    if (hist) for (int b = 0; b <= NC; b++) hist[b] = A.bp_hist[b];
    arena_to_host(A, G, *s);
}

// a whole rollout with the arena RESIDENT, as the HIP path keeps it (no hand-over in uu between steps): the host build's gym against the reference bit for bit
template <int NC>
static int gym_rollout_t(RlgpuArenaState* s, const GymConfig* cfg, const int32_t* actions, int steps, float* obs0, float* obs, float* rew, int32_t* done) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    const int P = cfg->one_team ? NC / 2 : NC; const size_t D = (size_t)obs_size<NC>(*cfg);
    if (obs0) gym_reset_env<NC>(A, G, *cfg, 0u, obs0, D, false);      // the episode's start on the uploaded state (no setter), first observation
    int t = 0;
    for (; t < steps; t++) {
        gym_step_env<NC>(A, G, *cfg, mv, table(), actions + (size_t)t * P, 0u, obs + (size_t)t * P * D, D, rew + (size_t)t * P, done + t, W);
        if (done[t]) { t++; break; }
    }
    arena_to_host(A, G, *s);
    return t;
}

extern "C" {

// steps taken (stops after the first done); obs0 [P][D] or null (no reset), obs [steps][P][D], rew [steps][P], done [steps]
int port_gym_rollout(RlgpuArenaState* s, const void* cfg, const int32_t* actions, int steps, float* obs0, float* obs, float* rew, int32_t* done) {
    const GymConfig* c = (const GymConfig*)cfg;
    if (s->num_cars == 2) return gym_rollout_t<2>(s, c, actions, steps, obs0, obs, rew, done);
    if (s->num_cars == 4) return gym_rollout_t<4>(s, c, actions, steps, obs0, obs, rew, done);
    return gym_rollout_t<6>(s, c, actions, steps, obs0, obs, rew, done);
}

// row width of the obs builder `cfg` describes (DefaultOBS, or DefaultOBSPadded when obs_max_players is set)
int port_obs_size(const void* cfg, int nc) { const GymConfig& g = *(const GymConfig*)cfg; return g.obs_max_players > 0 ? 51 + 38 * g.obs_max_players : 51 + 19 * (g.one_team ? nc / 2 : nc); }

void port_gym_reset(RlgpuArenaState* states, int n, const void* cfg, float* obs, int run_setter) {
    for (int e = 0; e < n; e++) {
        RlgpuArenaState* s = &states[e]; int nc = s->num_cars; int D = port_obs_size(cfg, nc);
        const int P = ((const GymConfig*)cfg)->one_team ? nc / 2 : nc;   // agent rows per env
        float* o = obs ? obs + (size_t)e * P * D : nullptr;
        if (nc == 2) gym_reset_t<2>(s, (const GymConfig*)cfg, e, o, run_setter);
        else if (nc == 4) gym_reset_t<4>(s, (const GymConfig*)cfg, e, o, run_setter);
        else gym_reset_t<6>(s, (const GymConfig*)cfg, e, o, run_setter);
    }
}
void port_gym_step_hist(RlgpuArenaState* states, int n, const void* cfg, const int32_t* actions, float* obs, float* rew, int32_t* done, uint16_t* hist /* [n][8] or null */) {
    for (int e = 0; e < n; e++) {
        uint16_t* h = hist ? hist + (size_t)e * 8 : nullptr;
        RlgpuArenaState* s = &states[e]; int nc = s->num_cars; int D = port_obs_size(cfg, nc);
        const int P = ((const GymConfig*)cfg)->one_team ? nc / 2 : nc;   // agent rows per env
        int32_t dn = 0;
        if (nc == 2) gym_step_t<2>(s, (const GymConfig*)cfg, e, actions + (size_t)e * P, obs + (size_t)e * P * D, rew + (size_t)e * P, &dn, h);
        else if (nc == 4) gym_step_t<4>(s, (const GymConfig*)cfg, e, actions + (size_t)e * P, obs + (size_t)e * P * D, rew + (size_t)e * P, &dn, h);
        else gym_step_t<6>(s, (const GymConfig*)cfg, e, actions + (size_t)e * P, obs + (size_t)e * P * D, rew + (size_t)e * P, &dn, h);
        for (int k = 0; k < P; k++) done[(size_t)e * P + k] = dn;
    }
}
void port_gym_step(RlgpuArenaState* states, int n, const void* cfg, const int32_t* actions, float* obs, float* rew, int32_t* done) {
    port_gym_step_hist(states, n, cfg, actions, obs, rew, done, nullptr);
}
int port_action_table(float* out) { memcpy(out, table(), sizeof(g_action_table)); return 90; }

// scalar CPU baseline ("port") of the collection hot loop: n envs stepped round-robin on n_threads threads
double port_bench_collect(int team_size, int n_envs, int n_threads, int steps, const void* cfg_v) {
    const GymConfig* cfg = (const GymConfig*)cfg_v;
    int nc = 2 * team_size; int D = port_obs_size(cfg_v, nc);
    std::vector<RlgpuArenaState> st(n_envs);
    for (auto& s : st) { memset(&s, 0, sizeof(s)); s.num_cars = nc; for (int k = 0; k < nc; k++) { s.cars[k].rot[0] = 1; s.cars[k].rot[4] = 1; s.cars[k].rot[8] = 1; } }
    std::vector<float> obs((size_t)n_envs * nc * D);
    port_gym_reset(st.data(), n_envs, cfg, obs.data(), 1);
    auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) {
        th.emplace_back([&, t]() {
            uint32_t rng = 12345u + 977u * t;
            std::vector<float> o((size_t)nc * D); float rew[6]; int32_t acts[6]; int32_t dn;
            for (int s = 0; s < steps; s++)
                for (int e = t; e < n_envs; e += n_threads) {
                    for (int k = 0; k < nc; k++) { rng = rng * 1664525u + 1013904223u; acts[k] = (rng >> 8) % 90; }
                    if (nc == 2) gym_step_t<2>(&st[e], cfg, e, acts, o.data(), rew, &dn);
                    else if (nc == 4) gym_step_t<4>(&st[e], cfg, e, acts, o.data(), rew, &dn);
                    else gym_step_t<6>(&st[e], cfg, e, acts, o.data(), rew, &dn);
                }
        });
    }
    for (auto& x : th) x.join();
    return std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
}

}  // extern "C"

// ---- debugging probe (tools/diff_ref_port.py): one tick, then the tick's contact list as the solver saw it -------------------
// per contact 16 floats: [0] body a (0 ball, 1+i car i, -1 world)  [1] body b  [2] row index  [3] special
//   [4..6] ra  [7..9] rb  [10..12] normal  [13] distance  [14] applied normal impulse  [15] friction
template <int NC>
static int debug_tick_t(RlgpuArenaState* s, float* out, int cap) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    TickEvents ev; ev.bump_mask = 0;
    arena_tick(A, mv, 0, 0, ev, W);
    arena_to_host(A, G, *s);
    int n = 0;
    for (int k = 0; k < W.L.n && n < cap; k++) {
        const Contact& c = W.L.c[W.cidx[k]];
        float* o = out + 16 * n++;
        o[0] = (float)c.a; o[1] = (float)c.b; o[2] = (float)W.nrow[k]; o[3] = c.special ? 1.f : 0.f;
        o[4] = c.ra.x; o[5] = c.ra.y; o[6] = c.ra.z; o[7] = c.rb.x; o[8] = c.rb.y; o[9] = c.rb.z;
        o[10] = c.n.x; o[11] = c.n.y; o[12] = c.n.z; o[13] = c.dist;
        o[14] = W.nrow[k] >= 0 ? W.R[W.nrow[k]].applied : 0.f; o[15] = contact_friction(c, A.mut);
    }
    return n;
}
// one tick, then car `slot`'s wheel scratch as the tick left it: per wheel 12 floats laid out like oracle/ref_driver.cpp:ref_debug_wheels
// (suspension length, 0, suspension relative velocity, clipped inverse contact dot, contact point, contact normal, hard point z, |impulse|)
template <int NC>
static void debug_wheels_t(RlgpuArenaState* s, int slot, float* out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    TickEvents ev; ev.bump_mask = 0;
    arena_tick(A, mv, 0, 0, ev, W);
    arena_to_host(A, G, *s);
    for (int w = 0; w < 4; w++) {
        const WheelTmp& wt = W.ctx[slot].w[w];
        float* o = out + 12 * w;
        o[0] = wt.susp_len; o[1] = 0.f; o[2] = wt.susp_rel_vel; o[3] = wt.clipped_inv;
        o[4] = wt.contact_point.x; o[5] = wt.contact_point.y; o[6] = wt.contact_point.z;
        o[7] = wt.contact_normal.x; o[8] = wt.contact_normal.y; o[9] = wt.contact_normal.z;
        o[10] = wt.hard_point.z; o[11] = len(wt.impulse);
    }
}
// the suspension rays car `slot` casts in the state's next tick: per wheel from3, to3 (Bullet units)
template <int NC>
static void debug_wheel_rays_t(const RlgpuArenaState* s, int slot, float* out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    const Car& car = A.cars[slot];
    const V3 wheel_dir = car.b.rot * v3(0, 0, -1);
    for (int i = 0; i < 4; i++) {
        const V3 from = (car.b.rot * wheel_conn(i)) + car.b.pos, to = from + wheel_dir * wheel_ray_len(i);
        float* o = out + 6 * i; o[0] = from.x; o[1] = from.y; o[2] = from.z; o[3] = to.x; o[4] = to.y; o[5] = to.z;
    }
}
extern "C" void port_debug_wheel_rays(const RlgpuArenaState* s, int slot, float* out) {
    if (s->num_cars == 2) debug_wheel_rays_t<2>(s, slot, out); else if (s->num_cars == 4) debug_wheel_rays_t<4>(s, slot, out); else debug_wheel_rays_t<6>(s, slot, out);
}
// the dynamic stage of a wheel ray (ball, other cars) on the state as it stands: out5 = kind, fraction, normal
template <int NC>
static void debug_ray_dynamic_t(const RlgpuArenaState* s, int self_car, const float* from3, const float* to3, float* out) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    RayHit best; best.kind = -1; best.frac = 1.f; best.normal = v3(0, 0, 0);
    ray_ball_and_cars(A, self_car, v3(from3[0], from3[1], from3[2]), v3(to3[0], to3[1], to3[2]), best);
    out[0] = (float)best.kind; out[1] = best.frac; out[2] = best.normal.x; out[3] = best.normal.y; out[4] = best.normal.z;
    for (int k = 0; k < NC; k++) out[5 + k] = (float)((A.cars[k].frozen ? 1 : 0) | ((A.cars[k].flags & CF_IS_DEMOED) ? 2 : 0));
}
extern "C" void port_debug_ray_dynamic(const RlgpuArenaState* s, int self_car, const float* from3, const float* to3, float* out) {
    if (s->num_cars == 2) debug_ray_dynamic_t<2>(s, self_car, from3, to3, out); else if (s->num_cars == 4) debug_ray_dynamic_t<4>(s, self_car, from3, to3, out); else debug_ray_dynamic_t<6>(s, self_car, from3, to3, out);
}
extern "C" void port_debug_wheels(RlgpuArenaState* s, int slot, float* out) {
    if (s->num_cars == 2) debug_wheels_t<2>(s, slot, out); else if (s->num_cars == 4) debug_wheels_t<4>(s, slot, out); else debug_wheels_t<6>(s, slot, out);
}
// friction rows of the tick's contacts in solver order: 8 floats = friction direction[3], applied friction impulse, friction coefficient, 0, applied normal impulse, 0
template <int NC>
static int debug_friction_t(RlgpuArenaState* s, float* out, int cap) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    PortWork<NC> W;
    TickEvents ev; ev.bump_mask = 0;
    arena_tick(A, mv, 0, 0, ev, W);
    int n = 0;
    for (int k = 0; k < W.L.n && n < cap; k++) {
        float* o = out + 8 * n++;
        const int fr = W.frow[k], nr = W.nrow[k];
        for (int q = 0; q < 8; q++) o[q] = 0.f;
        if (fr >= 0) { o[0] = W.R[fr].n1.x; o[1] = W.R[fr].n1.y; o[2] = W.R[fr].n1.z; o[3] = W.R[fr].applied; o[4] = W.R[fr].friction; }
        if (nr >= 0) o[6] = W.R[nr].applied;
    }
    return n;
}
extern "C" int port_debug_friction(RlgpuArenaState* s, float* out, int cap) {
    RlgpuArenaState c = *s;
    if (c.num_cars == 2) return debug_friction_t<2>(&c, out, cap);
    if (c.num_cars == 4) return debug_friction_t<4>(&c, out, cap);
    return debug_friction_t<6>(&c, out, cap);
}
extern "C" int port_debug_tick(RlgpuArenaState* s, float* out, int cap) {
    if (s->num_cars == 2) return debug_tick_t<2>(s, out, cap);
    if (s->num_cars == 4) return debug_tick_t<4>(s, out, cap);
    return debug_tick_t<6>(s, out, cap);
}
