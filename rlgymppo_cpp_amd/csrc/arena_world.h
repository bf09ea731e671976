// arena_world.h — static world (4 planes + BVH triangle mesh), suspension ray casts and the narrowphase that
// produces this tick's contact list.
//
// What it restates (reference = RocketSim 2.1.1's patched Bullet 3.24):
//   * world layout: floor z=0, ceiling z=2048, side walls x=+-4096 as planes (Arena.cpp:1060-1101) + trimesh
//   * ray tests: two-sided triangle/plane test with the ray-facing normal (btRaycastCallback.cpp:34-110),
//     closest hit wins (btDefaultVehicleRaycaster.cpp:32-52)
//   * sphere-plane / box-plane support-vertex test (btConvexPlaneCollisionAlgorithm.cpp:92-121)
//   * sphere-triangle (SphereTriangleDetector.cpp:139-241 with the embree closest-point routine :87-129)
//   * contact-added callback semantics (Arena.cpp:218-427)
//   * hitbox vs triangle / ball through Bullet's GJK pair detector, its penetration-depth solver (second GJK + EPA) and the wheel rays'
//     convex cast: arena_gjk.h, arena_epa.h, arena_simplex.h; car vs car through btBoxBoxDetector = ODE's dBoxBox2 (below); contact points
//     through btAdjustInternalEdgeContacts (adjust_internal_edge) -- all restated operation by operation and pinned bit for bit against
//     the reference's own routines (tests/golden/narrowphase_golden.npz)
// What it re-designs for a lane-per-env kernel:
//   * the mesh BVH is this repo's own 32-byte threaded AABB node layout over the reference's tree and triangle order (arena_mesh.cpp), with
//     a candidate / item queue so that the lanes of a wavefront share the narrowphase of their envs (CollideQueue below);
//   * SAT + face clipping (box_triangle, box_box) survive only as the stand-in behind RLG_EXPERIMENT_NO_EPA and for arenas too small for
//     the penetration-depth solver.
//
// Licence notes for the routines restated from third-party code (both permit use and redistribution of altered versions that are marked as such):
//   * Bullet Physics (btGjkPairDetector, btVoronoiSimplexSolver, btGjkEpa2, btSubsimplexConvexCast, btInternalEdgeUtility, btBoxBoxDetector's
//     driver, btPersistentManifold, btSequentialImpulseConstraintSolver): Copyright (c) 2003-2009 Erwin Coumans and contributors, zlib licence.
//   * ode_rect_quad / ode_cull_points / box_box_ode follow dBoxBox2 of the Open Dynamics Engine as shipped in Bullet's btBoxBoxDetector.cpp:
//     Copyright (c) 2001-2003 Russell L. Smith, BSD-style licence (either the LGPL or the BSD licence of ODE, at the user's option).
//   These files are altered restatements written for this repository, not the originals.
#pragma once
#include "arena_contact.h"
#include "arena_simplex.h"

namespace rlg {

// contact breaking thresholds = getAngularMotionDisc() * 0.02 of the smaller shape
// (btCollisionDispatcher.cpp:70-82, btCollisionShape.cpp:130-160; values verified against the compiled
// reference by oracle/ref_driver.cpp:ref_probe_thresholds)
constexpr float CBT_BALL = 0.0381f;   // (1.825 + 0.08) * 0.02
constexpr float CBT_CAR = 0.040624548f;  // compound(box+offset) bounding disc * 0.02 (probed: ref_probe_thresholds)

struct RayHit {
    int kind;  // -1 miss, 0 static world, 1 ball, 2+k car k
    float frac;
    V3 normal;
};


// ---- BVH access ---------------------------------------------------------------------------------------
// (MeshView travels by value: behind a reference it sits in the caller's stack frame = scratch memory on the device, and
// every node visit paid a dependent scratch round trip for `nodes_fast` before the node load itself.)
// one node = two 16-byte loads issued together (field-wise access made the compiler fetch the box first and the child / count
// words in a second, dependent round trip after the box test)
RLG_HD BvhNode load_node(const BvhNode* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    BvhNode n;
    n.minx = __uint_as_float(a.x); n.miny = __uint_as_float(a.y); n.minz = __uint_as_float(a.z); n.left_or_first = (int32_t)a.w;
    n.maxx = __uint_as_float(b.x); n.maxy = __uint_as_float(b.y); n.maxz = __uint_as_float(b.z); n.count_escape = b.w;
    return n;
#else
    return *p;
#endif
}
RLG_HD BvhNode mesh_node(MeshView m, int i) {
    if (i < m.n_fast) { RLG_ASSUME_LDS(*m.nodes_fast); return load_node(m.nodes_fast + i); }
    return load_node(m.nodes + i);
}

// does the box [lo,hi] touch any occupied grid cell?  (conservative: out-of-grid space counts as occupied)
RLG_HD bool mesh_maybe_near(MeshView m, V3 lo, V3 hi) {
    if (m.n_nodes <= 0) return false;
    if (!m.grid) return true;
    int x0 = (int)floorf((lo.x - GRID_MIN_X) * (1.f / GRID_CELL)), x1 = (int)floorf((hi.x - GRID_MIN_X) * (1.f / GRID_CELL));
    int y0 = (int)floorf((lo.y - GRID_MIN_Y) * (1.f / GRID_CELL)), y1 = (int)floorf((hi.y - GRID_MIN_Y) * (1.f / GRID_CELL));
    int z0 = (int)floorf((lo.z - GRID_MIN_Z) * (1.f / GRID_CELL)), z1 = (int)floorf((hi.z - GRID_MIN_Z) * (1.f / GRID_CELL));
    if (x0 < 0 || y0 < 0 || z0 < 0 || x1 >= GRID_X || y1 >= GRID_Y || z1 >= GRID_Z) return true;
    if ((x1 - x0) > 3 || (y1 - y0) > 3 || (z1 - z0) > 3) return true;
    RLG_ASSUME_LDS(*m.grid);
    for (int z = z0; z <= z1; z++)
        for (int y = y0; y <= y1; y++)
            for (int x = x0; x <= x1; x++) {
                int bit = (z * GRID_Y + y) * GRID_X + x;
                if ((m.grid[bit >> 5] >> (bit & 31)) & 1u) return true;
            }
    return false;
}
RLG_HD bool tri_aabb_overlap(const MeshTri& t, V3 lo, V3 hi) {
    if (fminf(t.v0x, fminf(t.v1x, t.v2x)) > hi.x || fmaxf(t.v0x, fmaxf(t.v1x, t.v2x)) < lo.x) return false;
    if (fminf(t.v0y, fminf(t.v1y, t.v2y)) > hi.y || fmaxf(t.v0y, fmaxf(t.v1y, t.v2y)) < lo.y) return false;
    if (fminf(t.v0z, fminf(t.v1z, t.v2z)) > hi.z || fmaxf(t.v0z, fmaxf(t.v1z, t.v2z)) < lo.z) return false;
    return true;
}

RLG_HD bool aabb_overlap(const BvhNode& n, V3 lo, V3 hi) {
    return !(n.minx > hi.x || n.maxx < lo.x || n.miny > hi.y || n.maxy < lo.y || n.minz > hi.z || n.maxz < lo.z);
}
// (the node's box grown by RAY_NODE_MARGIN: a ray test accepts hits up to 1e-4 x the triangle's height outside an edge -- 1.2 uu for the tallest wall triangle --
// and Bullet's own quantized boxes are a few quanta wider than the vertices'; the walk must not cull what ray_leaf_admits is there to decide)
constexpr float RAY_NODE_MARGIN = 0.05f;
RLG_HD bool ray_aabb(const BvhNode& n, V3 from, V3 inv_d, float tmax) {
    const float m = RAY_NODE_MARGIN;
    float t1 = ((n.minx - m) - from.x) * inv_d.x, t2 = ((n.maxx + m) - from.x) * inv_d.x;
    float tn = fminf(t1, t2), tf = fmaxf(t1, t2);
    t1 = ((n.miny - m) - from.y) * inv_d.y; t2 = ((n.maxy + m) - from.y) * inv_d.y;
    tn = fmaxf(tn, fminf(t1, t2)); tf = fminf(tf, fmaxf(t1, t2));
    t1 = ((n.minz - m) - from.z) * inv_d.z; t2 = ((n.maxz + m) - from.z) * inv_d.z;
    tn = fmaxf(tn, fminf(t1, t2)); tf = fminf(tf, fmaxf(t1, t2));
    return tf >= fmaxf(tn, 0.f) && tn <= tmax;
}

// btTriangleRaycastCallback::processTriangle (btRaycastCallback.cpp:34-110), flags = 0
// ---- narrowphase candidates (shared by the wheel rays and the contact narrowphase) ---------------------------
// keep the `cap` deepest candidates of one body-vs-world pair
// n_raw: the normal as the narrowphase reported it, BEFORE the internal-edge adjustment -- what the reference's contact-added callback hands
// to Arena::_BtCallback_OnCarWorldCollision (Arena.cpp:218-282: the callbacks run first, btAdjustInternalEdgeContacts last); n is the
// manifold point's normal after it.  Only the hitbox-triangle path fills n_raw.
// pa: the world point on A as the detector reported it (pb + n * dist with the UNadjusted normal, btManifoldResult.cpp:117): the manifold
// point's local point on A is taken from it before the callback moves pb along the adjusted normal (btInternalEdgeUtility.cpp: "reproject
// collision point along normal") -- rebuilding it from the adjusted pb and n is off by a rounding.  Filled where n_raw is, and by the ball's
// mesh contacts.
struct Cand { V3 pb; V3 n; float dist; V3 n_raw; V3 pa; };
template <int CAP>
RLG_HD void cand_add(Cand (&cs)[CAP], int& n, const Cand& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (CAP <= 4) {
        // every slot is addressed by a compile-time index (the insert position becomes a compare chain), so a caller-local array of
        // four candidates stays in registers: indexed by n / worst it lived in scratch memory (collide_body, once per body and tick)
        if (n < CAP) {
#pragma unroll
            for (int s = 0; s < CAP; s++) if (n == s) cs[s] = c;
            n++;
            return;
        }
        int worst = 0; float wd = cs[0].dist;
#pragma unroll
        for (int i = 1; i < CAP; i++) if (cs[i].dist > wd) { wd = cs[i].dist; worst = i; }
        if (c.dist < wd) {
#pragma unroll
            for (int s = 0; s < CAP; s++) if (worst == s) cs[s] = c;
        }
        return;
    }
#endif
    if (n < CAP) { cs[n++] = c; return; }
    int worst = 0;
    for (int i = 1; i < CAP; i++) if (cs[i].dist > cs[worst].dist) worst = i;
    if (c.dist < cs[worst].dist) cs[worst] = c;
}

// Three steps on the device, each on its own lane set:
//   1. collide_queue_body  (a lane per body)       walks the BVH nodes only and lists the triangles of every leaf it reaches
//                                                   as CANDIDATES -- no triangle is fetched during the walk;
//   2. collide_test_candidate (a lane per candidate) fetches the triangle and does the AABB test; the survivors are compacted,
//                                                   in candidate order, into ITEMS (rlgpu_env.hip does that with a ballot);
//   3. collide_run_item    (a lane per item)        runs the pair test and stores its candidates in the pool.
struct CollideItem {
    int16_t type, a;   // 0 ball-triangle, 1 car-triangle (a = car), 2 car-car (a = first car)
    int32_t ref;       // triangle index, or the second car
    int16_t off, n;    // result candidates: pool[off .. off+n)
};
#ifndef RLG_ITEM_CAP
#define RLG_ITEM_CAP 48   /* tests build a tiny queue to exercise the overflow fallback */
#endif
// measured over 614 K env-ticks of random play (tools: RLG_QSTAT hook): items <= 16 in 99.93 % of the ticks (max seen > 16), candidate slots <= 88,
// pool entries <= 12 -- the caps below leave the inline fallback to the truly pathological ticks
constexpr int ITEM_CAP = RLG_ITEM_CAP, POOL_CAP = 22;   // (pool entries of 52 bytes; measured fill <= 12)
constexpr int LEAF_SLOTS = 4;              // BVH leaves hold <= 4 triangles (arena_mesh.cpp); the device reserves a full block per leaf
constexpr uint32_t CAND_HOLE = 0xFFFFFFFFu;  // unused slot of such a block
#ifndef RLG_FRONTIER_CAP
#define RLG_FRONTIER_CAP 64   /* (128 until the per-file manifolds needed its LDS: the tessellated arena -- 10 k triangles -- never filled 64) */
#endif
constexpr int FRONTIER_CAP = RLG_FRONTIER_CAP;       // BVH nodes per level of the breadth-first walk (all bodies of an env together on the device)
#ifndef RLG_BODY_CAND
#define RLG_BODY_CAND 128   /* (96 until the tessellated arena -- 10 k triangles -- overflowed a car's region 5 times per 1000 env-ticks) */
#endif
static_assert(RLG_BODY_CAND % 4 == 0, "whole leaves");
constexpr int BALL_CAND = RLG_BODY_CAND, CAR_CAND = RLG_BODY_CAND, PAIR_SLOTS = 16;   // candidate slots: 24 leaves per body (measured: <= 22 leaves per env; 12 were not enough for the ball in the corners)
constexpr int CACHE_LEAVES = BALL_CAND / LEAF_SLOTS;   // leaves per body the device keeps
RLG_HD uint32_t pack_cand(int type, int a, int ref) { return ((uint32_t)type << 28) | ((uint32_t)a << 24) | (uint32_t)ref; }
// Candidate k of an env: the slots [region(b), region(b) + cand_count[b]) belong to body b (0 = ball, 1 + i = car i), the car pairs sit at
// PAIR_BASE.  The host build keeps every slot as a word.  The device build (RLG_QUEUE_LEAVES) keeps a body's LEAVES -- first triangle |
// count << 24, a block of LEAF_SLOTS slots each, unused slots are holes -- and derives slot k from them (queue_cand): a quarter of the LDS.
// Device member order: what the item phase still needs first (items, pool); the walk's frontier, the boxes and the leaves -- dead by then --
// last, so that together with what the solver rows add to the union (TickWork) they are one free stretch for the EPA arenas.
template <int NC>
struct CollideQueue {
    static constexpr int NB = NC + 1;
    static constexpr int PAIR_BASE = BALL_CAND + NC * CAR_CAND;
    int n_items, n_pool, overflow, n_pairs;
    uint16_t cand_count[8];
#ifdef RLG_QUEUE_LEAVES
    CollideItem items[ITEM_CAP];
    Cand pool[POOL_CAP];
    uint32_t frontier[2][FRONTIER_CAP];   // breadth-first BVH walk: (body << 16 | node) of the current and of the next level
    V3 box_lo[NB], box_hi[NB];            // the bodies' query boxes during the device walk
    uint32_t leaf[NB][CACHE_LEAVES];      // this tick's leaves per body, ascending first triangle (= the reference's visiting order)
    uint32_t pair[PAIR_SLOTS];            // type << 28 | a << 24 | b
#else
    uint32_t frontier[2][FRONTIER_CAP];
    V3 box_lo[NB], box_hi[NB];
    uint32_t cand[PAIR_BASE + PAIR_SLOTS];   // type << 28 | a << 24 | ref
    CollideItem items[ITEM_CAP];
    Cand pool[POOL_CAP];
#endif
    static RLG_HD int region(int body) { return body == 0 ? 0 : BALL_CAND + (body - 1) * CAR_CAND; }
    static RLG_HD int region_cap(int body) { return body == 0 ? BALL_CAND : CAR_CAND; }
};
template <int NC>
RLG_HD uint32_t queue_cand(const CollideQueue<NC>& Q, int k) {
#ifdef RLG_QUEUE_LEAVES
    if (k >= CollideQueue<NC>::PAIR_BASE) return Q.pair[k - CollideQueue<NC>::PAIR_BASE];
    const int body = k < BALL_CAND ? 0 : 1 + (k - BALL_CAND) / CAR_CAND;
    const int slot = k - CollideQueue<NC>::region(body);
    const uint32_t lf = Q.leaf[body][slot / LEAF_SLOTS];
    const int q = slot % LEAF_SLOTS, first = (int)(lf & 0xFFFFFFu), cnt = (int)(lf >> 24);
    return q < cnt ? pack_cand(body == 0 ? 0 : 1, body == 0 ? 0 : body - 1, first + q) : CAND_HOLE;
#else
    return Q.cand[k];
#endif
}
RLG_HD CollideItem unpack_cand(uint32_t c) {
    CollideItem it; it.type = (int16_t)(c >> 28); it.a = (int16_t)((c >> 24) & 15u); it.ref = (int32_t)(c & 0xFFFFFFu); it.off = 0; it.n = 0;
    return it;
}

RLG_HD int fetch_add(int& x, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    // lanes of one wavefront append to the same queue; the queue lives in LDS (ds_add_rtn instead of a flat atomic)
    return __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)&x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    int o = x; x += v; return o;
#endif
}

template <int NC>
RLG_HD void queue_candidates(CollideQueue<NC>& Q, int body, int first, int cnt) {   // host form of the walk (the device's: rlgpu_env.hip build_candidates_wave)
    int k = Q.cand_count[body];
#ifdef RLG_QUEUE_LEAVES
    if (RLG_UNLIKELY(k / LEAF_SLOTS >= CACHE_LEAVES)) { Q.overflow = 1; return; }
    Q.leaf[body][k / LEAF_SLOTS] = (uint32_t)first | ((uint32_t)cnt << 24);
    Q.cand_count[body] = (uint16_t)(k + LEAF_SLOTS);
#else
    if (RLG_UNLIKELY(k + cnt > CollideQueue<NC>::region_cap(body))) { Q.overflow = 1; return; }
    Q.cand_count[body] = (uint16_t)(k + cnt);
    const int type = body == 0 ? 0 : 1, a = body == 0 ? 0 : body - 1;
    for (int q = 0; q < cnt; q++) Q.cand[CollideQueue<NC>::region(body) + k + q] = pack_cand(type, a, first + q);
#endif
}
template <int NC>
RLG_HD void queue_pair(CollideQueue<NC>& Q, int ia, int ib) {
    if (RLG_UNLIKELY(Q.n_pairs >= PAIR_SLOTS)) { Q.overflow = 1; return; }
#ifdef RLG_QUEUE_LEAVES
    Q.pair[Q.n_pairs++] = pack_cand(2, ia, ib);
#else
    Q.cand[CollideQueue<NC>::PAIR_BASE + Q.n_pairs++] = pack_cand(2, ia, ib);
#endif
}

// every BVH leaf whose box overlaps [lo,hi], in walk order: f(first triangle, count)
template <class F>
RLG_HD void mesh_query_leaves(MeshView mesh, V3 lo, V3 hi, F&& f) {
    if (!mesh_maybe_near(mesh, lo, hi)) return;
    uint32_t i = 0;
    while (i != BVH_END) {
        BvhNode nd = mesh_node(mesh, (int)i);
        uint32_t next = node_escape(nd);
        if (aabb_overlap(nd, lo, hi)) {
            const int cnt = node_count(nd);
            if (cnt > 0) f(nd.left_or_first, cnt);
            else next = (uint32_t)nd.left_or_first;   // left child first: leaves come out in ascending first-triangle order = the reference's visiting order
        }
        i = next;
    }
}

RLG_HD void ball_query_aabb(V3 bp, V3& lo, V3& hi) {
    const float r = K::BALL_RADIUS * UU2BT;
    float bext = r + 0.08f + 0.04f;  // sphere AABB (+0.08 patch, btSphereShape.cpp:55) grown by the trimesh margin
    lo = bp - v3(bext, bext, bext); hi = bp + v3(bext, bext, bext);
}
RLG_HD void car_query_aabb(const Car& car, V3& bc, V3& lo, V3& hi) {
    V3 h = hitbox_half();
    bc = car.b.pos + car.b.rot * hitbox_off();
    M3 absR = m3_rows(v3(fabsf(car.b.rot.r0.x), fabsf(car.b.rot.r0.y), fabsf(car.b.rot.r0.z)),
                      v3(fabsf(car.b.rot.r1.x), fabsf(car.b.rot.r1.y), fabsf(car.b.rot.r1.z)),
                      v3(fabsf(car.b.rot.r2.x), fabsf(car.b.rot.r2.y), fabsf(car.b.rot.r2.z)));
    // btConvexTriangleCallback::setTimeStepAndCounters (btConvexConcaveCollisionAlgorithm.cpp:167-188): the box's own AABB (margin included)
    // grown by the trimesh's collision margin, which is 0 (btConcaveShape.cpp:21) -- a triangle the hitbox is merely NEAR is never tested
    V3 ext = absR * h;
    lo = bc - ext; hi = bc + ext;
}
// CF_NO_CONTACT_RESPONSE + DISABLE_SIMULATION are set by the NEXT pre-tick of a demoed car (Car.cpp:69-80): `frozen` = demoed at tick start.
// A car demolished by this tick's callback (Car::Demolish only sets the state flag) still collides and integrates until the tick ends.
RLG_HD bool car_collides(const Car& car) { return !car.frozen; }
template <int NC>
RLG_HD bool cars_maybe_touch(const Arena<NC>& A, int ia, int ib) {
    const Car& ca = A.cars[ia]; const Car& cb = A.cars[ib];
    V3 h = hitbox_half();
    V3 cca = ca.b.pos + ca.b.rot * hitbox_off(), ccb = cb.b.pos + cb.b.rot * hitbox_off();
    float rad = len(h);
    return len2(cca - ccb) <= (2 * rad) * (2 * rad);
}

// the box the car's candidates are collected for: hitbox query box united with its four suspension rays (the rays are cast
// from the same pose, before anything moves, so ONE candidate list per car and tick serves both)
RLG_HD float wheel_ray_len(int i) { return wheel_rest(i) + wheel_travel() + wheel_radius(i) - K::SUSPENSION_SUBTRACTION; }
RLG_HD void car_wide_aabb(const Car& car, V3& lo, V3& hi) {
    V3 bc;
    car_query_aabb(car, bc, lo, hi);
    V3 wheel_dir = car.b.rot * v3(0, 0, -1);
    for (int i = 0; i < 4; i++) {
        V3 from = (car.b.rot * wheel_conn(i)) + car.b.pos, to = from + wheel_dir * wheel_ray_len(i);
        lo = v3(fminf(lo.x, fminf(from.x, to.x)), fminf(lo.y, fminf(from.y, to.y)), fminf(lo.z, fminf(from.z, to.z)));
        hi = v3(fmaxf(hi.x, fmaxf(from.x, to.x)), fmaxf(hi.y, fmaxf(from.y, to.y)), fmaxf(hi.z, fmaxf(from.z, to.z)));
    }
}
// which bodies get candidates this tick, and for which box.  A car that is frozen but not demoed (respawned this tick) still
// casts wheel rays, so it keeps its list; the narrowphase ignores it (car_collides).
template <int NC>
RLG_HD bool body_query_box(const Arena<NC>& A, int body, bool ball_asleep, V3& lo, V3& hi) {
    if (body == 0) {
        if (ball_asleep) return false;
        ball_query_aabb(A.ball.b.pos, lo, hi);
        return true;
    }
    const Car& car = A.cars[body - 1];
    if (car.flags & CF_IS_DEMOED) return false;
    car_wide_aabb(car, lo, hi);
    return true;
}

// every mesh triangle whose AABB overlaps [lo,hi] (TestTriangleAgainstAabb2, btConvexConcaveCollisionAlgorithm.cpp:75), in the reference's
// visiting order, for the inline narrowphase (queue overflow fallback): falling back changes nothing but speed.
template <class F>
RLG_HD void mesh_query(MeshView mesh, V3 lo, V3 hi, F&& f) {
    mesh_query_leaves(mesh, lo, hi, [&](int first, int cnt) {
        for (int k = 0; k < cnt; k++)
            if (tri_aabb_overlap(mesh.tris[first + k], lo, hi)) f(first + k);
    });
}

// all candidates of one env for this tick (host form; called before the wheel rays): per body the overlapping leaves in ascending
// first-triangle order, which is the reference's visiting order (arena_mesh.cpp).  The device walks all bodies of an env in ONE
// breadth-first pass, keeps the leaves over several ticks and sorts them (rlgpu_env.hip:build_candidates_wave): same sequence per body.
template <int NC>
RLG_HD void collide_build_candidates(const Arena<NC>& A, MeshView mesh, bool ball_asleep, CollideQueue<NC>& Q) {
    Q.n_items = 0; Q.n_pool = 0; Q.overflow = 0; Q.n_pairs = 0;
    if (mesh.n_nodes > 65535) Q.overflow = 1;   // frontier entries carry 16-bit node ids: bigger trees use the inline walk
    for (int body = 0; body <= NC; body++) {
        Q.cand_count[body] = 0;
        V3 lo, hi;
        if (Q.overflow || !body_query_box(A, body, ball_asleep, lo, hi)) continue;
        mesh_query_leaves(mesh, lo, hi, [&](int first, int cnt) { queue_candidates(Q, body, first, cnt); });
    }
    for (int ci = 0; ci < NC; ci++)
        for (int ib = ci + 1; ib < NC; ib++)
            if (car_collides(A.cars[ci]) && car_collides(A.cars[ib]) && cars_maybe_touch(A, ci, ib)) queue_pair(Q, ci, ib);
}

// Bullet reaches a mesh triangle with a ray only through the triangle's LEAF of the mesh object's quantized tree (btBvhTriangleMeshShape::performRaycast ->
// btQuantizedBvh::walkStacklessQuantizedTreeAgainstRay, btQuantizedBvh.cpp:531-650): the quantized box of the ray has to overlap the leaf's quantized box, and the
// ray the dequantized one (btRayAabb2, btAabbUtil2.h:90-130).  A hit INSIDE its triangle passes both by construction; btTriangleRaycastCallback also accepts a
// hit up to 1e-4 |n|^2 OUTSIDE an edge (edge_tolerance), and such a hit counts only if the walk got to the leaf -- next to an open edge (a goal mouth's post) it
// may not have, and the ray goes on to what is behind.  Found in round 6 by the mutator fixture's `M2/cannon` tape (a wheel ray 0.19 uu beside the back wall's
// edge at the goal post; before, the tolerance alone decided).  `frame`: the object's bvhAabbMin, bvhAabbMax, bvhQuantization (arena_mesh.cpp QBvhFrame).
RLG_HD_COLD bool ray_leaf_admits(const float* frame, V3 v0, V3 v1, V3 v2, V3 from, V3 to) {
    const float fmn[3] = {frame[0], frame[1], frame[2]}, fmx[3] = {frame[3], frame[4], frame[5]}, fq[3] = {frame[6], frame[7], frame[8]};
    // the leaf's box as btOptimizedBvh::build made it (btOptimizedBvh.cpp:95-123): the vertices' box, a flat side widened, quantized outwards
    float mn[3] = {fminf(fminf(v0.x, v1.x), v2.x), fminf(fminf(v0.y, v1.y), v2.y), fminf(fminf(v0.z, v1.z), v2.z)};
    float mx[3] = {fmaxf(fmaxf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.z, v1.z), v2.z)};
    uint16_t lmn[3], lmx[3], rmn[3], rmx[3];
    const float rlo[3] = {fminf(from.x, to.x), fminf(from.y, to.y), fminf(from.z, to.z)}, rhi[3] = {fmaxf(from.x, to.x), fmaxf(from.y, to.y), fmaxf(from.z, to.z)};
    for (int a = 0; a < 3; a++) {
        if (mx[a] - mn[a] < 0.002f) { mx[a] = mx[a] + 0.001f; mn[a] = mn[a] - 0.001f; }
        lmn[a] = (uint16_t)(((uint16_t)((mn[a] - fmn[a]) * fq[a])) & 0xfffe);                 // quantize(.., isMax = 0 / 1), btQuantizedBvh.h:331-358
        lmx[a] = (uint16_t)(((uint16_t)((mx[a] - fmn[a]) * fq[a] + 1.f)) | 1);
        const float clo = fminf(fmaxf(rlo[a], fmn[a]), fmx[a]), chi = fminf(fmaxf(rhi[a], fmn[a]), fmx[a]);   // quantizeWithClamp
        rmn[a] = (uint16_t)(((uint16_t)((clo - fmn[a]) * fq[a])) & 0xfffe);
        rmx[a] = (uint16_t)(((uint16_t)((chi - fmn[a]) * fq[a] + 1.f)) | 1);
    }
    for (int a = 0; a < 3; a++) if (!(rmn[a] <= lmx[a] && rmx[a] >= lmn[a])) return false;      // testQuantizedAabbAgainstQuantizedAabb
    float b0[3], b1[3];
    for (int a = 0; a < 3; a++) { float v = (float)lmn[a] / fq[a]; v += fmn[a]; b0[a] = v; float w = (float)lmx[a] / fq[a]; w += fmn[a]; b1[a] = w; }   // unQuantize
    const V3 seg = to - from;
    const V3 dir = safe_normalized(seg);
    const float lambda_max = dot(dir, seg);
    const float inv[3] = {dir.x == 0.f ? 1e18f : 1.0f / dir.x, dir.y == 0.f ? 1e18f : 1.0f / dir.y, dir.z == 0.f ? 1e18f : 1.0f / dir.z};   // BT_LARGE_FLOAT
    const float src[3] = {from.x, from.y, from.z};
    // btRayAabb2 with lambda_min = 0
    float tmin = ((inv[0] < 0.f ? b1[0] : b0[0]) - src[0]) * inv[0], tmax = ((inv[0] < 0.f ? b0[0] : b1[0]) - src[0]) * inv[0];
    const float tymin = ((inv[1] < 0.f ? b1[1] : b0[1]) - src[1]) * inv[1], tymax = ((inv[1] < 0.f ? b0[1] : b1[1]) - src[1]) * inv[1];
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((inv[2] < 0.f ? b1[2] : b0[2]) - src[2]) * inv[2], tzmax = ((inv[2] < 0.f ? b0[2] : b1[2]) - src[2]) * inv[2];
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < lambda_max) && (tmax > 0.f);
}
// the frame of mesh object `obj`: behind the broadphase blob (arena_mesh.cpp)
RLG_HD const float* mesh_leaf_frame(const uint32_t* bp, uint32_t obj) {
    const uint32_t n_obj = bp[0];
    return reinterpret_cast<const float*>(bp + 1 + (size_t)n_obj * 6 + (size_t)(BP_CELLS_X * BP_CELLS_Y * BP_CELLS_Z) + (size_t)obj * 9);
}

// bp / tri: a MESH triangle's ray test (null for the static planes' two triangles, which no tree stands in front of)
RLG_HD void ray_triangle(V3 v0, V3 v1, V3 v2, V3 from, V3 to, RayHit& best, const uint32_t* bp = nullptr, const MeshTri* tri = nullptr) {
    V3 v10 = v1 - v0, v20 = v2 - v0;
    V3 tn = cross(v10, v20);
    float dist = dot(v0, tn);
    float da = dot(tn, from) - dist, db = dot(tn, to) - dist;
    if (da * db >= 0.f) return;
    float proj = da - db;
    float d = da / proj;
    if (d < best.frac) {
        float edge_tol = len2(tn) * -0.0001f;
        float s = 1.f - d;
        V3 p = v3(s * from.x + d * to.x, s * from.y + d * to.y, s * from.z + d * to.z);
        V3 v0p = v0 - p, v1p = v1 - p;
        const float e0 = dot(cross(v0p, v1p), tn);
        if (e0 >= edge_tol) {
            V3 v2p = v2 - p;
            const float e1 = dot(cross(v1p, v2p), tn);
            if (e1 >= edge_tol) {
                const float e2 = dot(cross(v2p, v0p), tn);
                if (e2 >= edge_tol) {
                    // accepted by the tolerance only, i.e. outside the triangle: did Bullet's tree walk get to this leaf?
#ifndef RLG_TEST_NO_LEAF_ADMISSION   /* test builds only (tests/golden/make_edge_golden.py checks that its fixture fails without the rule) */
                    if (RLG_UNLIKELY(bp != nullptr && (e0 < 0.f || e1 < 0.f || e2 < 0.f)) && !ray_leaf_admits(mesh_leaf_frame(bp, tri->obj), v0, v1, v2, from, to)) return;
#endif
                    V3 nn = normalized(tn);
                    best.frac = d; best.kind = 0;
                    best.normal = (da <= 0.f) ? -nn : nn;
                }
            }
        }
    }
}

// the four world planes: n.x = d
RLG_HD void world_plane(int i, V3& n, float& d) {
    const float ex = K::ARENA_EXTENT_X * UU2BT, h = K::ARENA_HEIGHT * UU2BT;
    if (i == 0) { n = v3(0, 0, 1); d = 0.f; }
    else if (i == 1) { n = v3(0, 0, -1); d = -h; }
    else if (i == 2) { n = v3(1, 0, 0); d = -ex; }
    else { n = v3(-1, 0, 0); d = -ex; }
}

// the same four planes as the reference builds them (Arena.cpp:1060-1101): a btStaticPlaneShape (normal n, constant 0) on a static
// body whose world transform is a pure translation `origin` -- the convex-plane test runs in that body's frame
RLG_HD void world_plane_body(int i, V3& n, V3& origin) {
    const float ex = K::ARENA_EXTENT_X, h = K::ARENA_HEIGHT;
    // btStaticPlaneShape keeps planeNormal.normalized() (btStaticPlaneShape.cpp:19): through the reference's rsqrtss-based normalize the
    // unit axis becomes 0.99999994 of itself, and that is the normal its contact and ray code work with
    const float u = normalized(v3(0, 0, 1)).z;
    if (i == 0) { n = v3(0, 0, u); origin = v3(0, 0, 0); }
    else if (i == 1) { n = v3(0, 0, -u); origin = v3(0.f * UU2BT, 0.f * UU2BT, h * UU2BT); }
    else if (i == 2) { n = v3(u, 0, 0); origin = v3(-ex * UU2BT, 0.f * UU2BT, (h / 2) * UU2BT); }
    else { n = v3(-u, 0, 0); origin = v3(ex * UU2BT, 0.f * UU2BT, (h / 2) * UU2BT); }
}

// the box of plane body i: RocketSim's btStaticPlaneShape::getAabb (btStaticPlaneShape.cpp:40-54) is a half space that ends 0.2 in front
// of the plane, so a body is only paired with a plane it is within reach of.  `grow` = 0: the SHAPE's box, which a compound's child
// test reads (btCompoundCollisionAlgorithm.cpp:127-141); `grow` = gContactBreakingThreshold = 0.02: the PROXY's box -- updateAabbs visits
// the static bodies too (m_forceUpdateAllAabbs, btCollisionWorld.cpp:62,143-193), so from the first tick on the broadphase holds the grown
// box like every other (the live reference's floor proxy ends at z = 0.22).  A ball whose own box ends between 0.2 and 0.22 has an (empty)
// manifold with the floor, and an empty manifold still takes a place in the island sort (round 6, tools/live_gym_hip.py).
RLG_HD void world_plane_aabb(int i, V3& lo, V3& hi, float grow = 0.f) {
    const float L = 1e18f;
    V3 n, o; world_plane_body(i, n, o);
    lo = v3(-L, -L, -L); hi = v3(L, L, L);
    if (i == 0) hi.z = (o.z + (0.f + 0.2f)) + grow;
    else if (i == 1) lo.z = (o.z + (0.f - 0.2f)) - grow;
    else if (i == 2) hi.x = (o.x + (0.f + 0.2f)) + grow;
    else lo.x = (o.x + (0.f - 0.2f)) - grow;
}

// What a contact point looks like by the time the solver reads it: btManifoldResult::addContactPoint stores the point in both
// bodies' local frames (btManifoldResult.cpp:128-150) and every narrowphase algorithm ends with refreshContactPoints, which rebuilds
// the world positions and the distance from those local points (btPersistentManifold.cpp:245-256).  `b_origin`: world origin of
// body B when it is a static plane body (pure translation), zero otherwise.  Out: ra = positionWorldOnA - origin of A, pb_w.
RLG_HD void manifold_point_refresh(const Body& a, V3 pa_w, V3 pb_w, V3 n, V3 b_origin, V3& ra, V3& pb_out, float& dist) {
    V3 la = tmul(a.rot, pa_w - a.pos);             // btTransform::invXform
    V3 wa = (a.rot * la) + a.pos;                   // trA(localPointA)
    V3 wb = (pb_w - b_origin) + b_origin;           // static body: identity basis
    dist = dot(wa - wb, n);
    ra = wa - a.pos; pb_out = wb;
}
// both bodies dynamic
RLG_HD void manifold_point_refresh2(const Body& a, const Body& b, V3 pa_w, V3 pb_w, V3 n, V3& ra, V3& rb, float& dist) {
    V3 la = tmul(a.rot, pa_w - a.pos), lb = tmul(b.rot, pb_w - b.pos);
    V3 wa = (a.rot * la) + a.pos, wb = (b.rot * lb) + b.pos;
    dist = dot(wa - wb, n);
    ra = wa - a.pos; rb = wb - b.pos;
}

// A suspension ray is cast in three stages (planes | mesh | ball and cars) so that the mesh stage can run as one lane per
// (ray, candidate triangle) pair on the device; every later stage only accepts strictly closer hits, as one loop would.
// A suspension ray against the four arena planes, the way the reference gets there: every car's own box marks the suspension grid
// cells around it as "dynamic" (Arena.cpp:733-748, SuspensionCollisionGrid.cpp:185-205), so CastSuspensionRay never takes its analytic
// shortcut for a car's wheel (:124-134) and the ray goes through btCollisionWorld::rayTest.  There a btStaticPlaneShape is a concave shape:
// the ray, moved into the plane body's frame (a pure translation), is tested against the TWO TRIANGLES btStaticPlaneShape::processAllTriangles
// spans over the ray's own box (btStaticPlaneShape.cpp:56-82) with btTriangleRaycastCallback (= ray_triangle above).  Same hits as the
// analytic intersection, but the hit fraction comes out of the triangles' cross products -- a few ulps that every suspension length
// and force of a car on the ground inherits.  (A plane the ray stays strictly on one side of cannot be hit by its triangles either: the callback's
// first test is the same sign test, with the triangle normal c * n.)
RLG_HD RayHit ray_planes(V3 from, V3 to) {
    RayHit best; best.kind = -1; best.frac = 1.0f; best.normal = v3(0, 0, 0);
    for (int i = 0; i < 4; i++) {
        V3 n, origin; world_plane_body(i, n, origin);
        const V3 lf = from - origin, lt = to - origin;
        const float da = dot(n, lf), db = dot(n, lt);
        // (strictly: an end point ON the plane -- product 0 -- is for the triangles' own sign test to decide: their normal and offset come out of rounded vertices, and
        // Bullet does report the hit at fraction ~1 that the analytic test would drop.  Round 6, tools/random_tapes.py ... walls, seed 9044 tick 360.)
#ifdef RLG_TEST_ANALYTIC_PLANE_SIGN    /* test builds only, as above */
        if (da * db >= 0.f) continue;
#else
        if (da * db > 0.f) continue;
#endif
        const V3 bmin = v3(fminf(lf.x, lt.x), fminf(lf.y, lt.y), fminf(lf.z, lt.z)), bmax = v3(fmaxf(lf.x, lt.x), fmaxf(lf.y, lt.y), fmaxf(lf.z, lt.z));
        const V3 half = (bmax - bmin) * 0.5f;
        const float radius = len(half);
        const V3 center = (bmax + bmin) * 0.5f;
        V3 t0, t1; plane_space1(n, t0, t1);
        const V3 pc = center - n * (dot(n, center) - 0.f);
        const V3 a0 = t0 * radius, a1 = t1 * radius;
        const V3 ppp = (pc + a0) + a1, ppm = (pc + a0) - a1, pmm = (pc - a0) - a1, pmp = (pc - a0) + a1;
        ray_triangle(ppp, ppm, pmm, lf, lt, best);
        ray_triangle(pmm, pmp, ppp, lf, lt, best);
    }
    return best;
}

// mesh stage, one (ray, triangle) pair: hit closer than `bound`?  (same test as ray_triangle)
RLG_HD bool ray_triangle_pair(const uint32_t* bp, const MeshTri& t, V3 from, V3 to, float bound, float& d_out) {
    RayHit probe; probe.kind = -1; probe.frac = bound; probe.normal = v3(0, 0, 0);
    ray_triangle(v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z), from, to, probe, bp, &t);
    d_out = probe.frac;
    return probe.kind == 0;
}
// the winner of the pairs: key = frac bits << 32 | candidate slot (smaller = closer, then earlier), ~0 = no hit
constexpr unsigned long long RAY_NO_HIT = ~0ull;
RLG_HD unsigned long long ray_key(float d, int slot) { return ((unsigned long long)f2u(d) << 32) | (unsigned long long)(uint32_t)slot; }
RLG_HD void ray_key_min(unsigned long long& key, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_min((__attribute__((address_space(3))) unsigned long long*)&key, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    if (v < key) key = v;
#endif
}
template <class QT>
RLG_HD void ray_apply_mesh_key(MeshView mesh, const QT& Q, unsigned long long key, V3 from, V3 to, RayHit& best) {
    if (key == RAY_NO_HIT) return;
    const MeshTri& t = mesh.tris[unpack_cand(queue_cand(Q, (int)(uint32_t)key)).ref];
    RayHit h; h.kind = -1; h.frac = best.frac; h.normal = v3(0, 0, 0);
    ray_triangle(v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z), from, to, h, mesh.bp, &t);
    if (h.kind == 0) best = h;   // recomputed from the winning triangle: same frac, and its ray-facing normal
}
// mesh stage without a candidate list (queue overflow): depth-first BVH walk
RLG_HD_COLD void ray_mesh_walk(MeshView mesh, V3 from, V3 to, RayHit& best) {
    if (!mesh_maybe_near(mesh, v3(fminf(from.x, to.x), fminf(from.y, to.y), fminf(from.z, to.z)), v3(fmaxf(from.x, to.x), fmaxf(from.y, to.y), fmaxf(from.z, to.z)))) return;
    V3 dvec = to - from;
    V3 inv_d = v3(1.f / dvec.x, 1.f / dvec.y, 1.f / dvec.z);
    uint32_t i = 0;
    while (i != BVH_END) {
        BvhNode nd = mesh_node(mesh, (int)i);
        uint32_t next = node_escape(nd);
        if (ray_aabb(nd, from, inv_d, best.frac)) {
            const int cnt = node_count(nd);
            if (cnt > 0) {
                for (int k = 0; k < cnt; k++) {
                    const MeshTri& t = mesh.tris[nd.left_or_first + k];
                    ray_triangle(v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z), from, to, best, mesh.bp, &t);
                }
            } else next = (uint32_t)nd.left_or_first;   // left child first: leaves come out in ascending first-triangle order = the reference's visiting order
        }
        i = next;
    }
}

// One dynamic object in a wheel ray's way, as btCollisionWorld::rayTestSingleInternal handles a convex shape (btCollisionWorld.cpp:267-310):
// the subsimplex cast of a point, the reported normal normalised, kept when its fraction is below the closest so far.
RLG_HD bool ray_convex_hit(V3 from, V3 to, const M3& Rb, V3 ob, V3 half, float radius, int kind, RayHit& best) {
    float frac; V3 n;
    if (!ray_convex_cast(from, to, Rb, ob, half, radius, frac, n)) return false;
    if (!(len2(n) > 0.0001f)) return false;
    if (!(frac < best.frac)) return false;
    best.frac = frac; best.kind = kind; best.normal = normalized(n);
    return true;
}
// the ray's and an object's boxes must come within 0.05 of each other before the cast is worth its GJK iterations (not in the reference's
// result: the cast reports a hit only once the point is within sqrt(1e-4) = 0.01 of the shape, so a ray farther than that from the
// object's box ends with lambda > 1 or VdotR >= 0)
RLG_HD bool ray_box_near(V3 from, V3 to, V3 lo, V3 hi) {
    const V3 m = v3(0.05f, 0.05f, 0.05f); lo = lo - m; hi = hi + m;
    return !(fminf(from.x, to.x) > hi.x || fmaxf(from.x, to.x) < lo.x || fminf(from.y, to.y) > hi.y || fmaxf(from.y, to.y) < lo.y || fminf(from.z, to.z) > hi.z || fmaxf(from.z, to.z) < lo.z);
}
// The rigid body of a demolished car stays in the world, disabled, where the demolition tick left it (Car.cpp:69-80) -- and its basis is
// NOT the one the car's state reports: Car::_PostTickUpdate stops copying the rotation once the car is demoed (Car.cpp:135-138), so the
// state keeps the basis from before the demolition tick while the body has turned through it (a supersonic hit: several degrees).  The body's
// own basis is kept in the slot of the world inverse inertia, which a disabled body has no use for (written by solver_finish, restored by
// the respawn; resident layout: arena_io.h).
RLG_HD const M3& car_ghost_rot(const Car& car) { return car.b.inv_inertia_w; }
// (after car_tick_begin) this is the wreck's first pre-tick: Car::Demolish set the timer, one tick has been taken off it
RLG_HD bool car_demolished_last_tick(const Car& car, float respawn_delay) { return (car.flags & CF_IS_DEMOED) && car.demo_respawn_timer == fmaxf(respawn_delay - TICK_DT, 0.f); }
template <int NC>
RLG_HD int car_rank_of(const Arena<NC>& A, int slot) { for (int k = 0; k < NC; k++) if (car_at_rank(A, k) == slot) return k; return slot; }
// last stage: the dynamic objects -- the ball (a btSphereShape), then the other cars' hitbox children (btBoxShape), each through the convex
// cast.  (The ball's rotation is not part of the state this build keeps; its basis is taken as the identity: the support point of a sphere
// does not depend on it beyond rounding.)
// Which dynamic objects a wheel ray is cast against at all.  The reference's broadphase hands a short ray EVERY dynamic proxy on the list of the cell its origin
// lies in (btRSBroadphase.cpp:326-337) -- a proxy is on the lists of the 27 cells around the one its box's minimum corner was filed under at its last setAabb
// (:183-200), which is what Arena::bp_hist remembers -- and btCollisionWorld::rayTestSingle runs the convex cast without looking at a box first.  That matters
// because btSubsimplexConvexCast returns "hit" when its 32 iterations run out: for a ray that passes a tumbling car at some distance it now and then reports a hit
// 20 - 30 uu OUTSIDE the box (tools/random_tapes.py ... aerial, seeds 30032 / 30118 / 30125, round 6: a wheel "standing" on a car it does not touch), and a test of
// the ray's box against the car's throws exactly those away.  Reproducing them costs 2.7 % of configs[1]'s throughput (4.8 % in 3v3: in a chase the other car and
// the ball are on the list most of the time, and every wheel then pays at least the cast's first pass), for an artefact -- a car pushed by a car it does not
// touch -- that showed up 3 times in 120 000 ticks of cars tumbling within 900 uu of one another and never in 2 M ticks of anything else: it is a per-env switch,
// RLGPU_MUT_RAY_PROXY_LISTS in RlgpuMutators::flags, OFF by default (the box test alone decides, as before), ON in the fixtures that pin it.  A body without a filed cell (bp_hist 0: a fresh arena, a state
// uploaded without its hidden block) keeps the box test.  (A ray whose box comes near an object is always in a listed cell: the lists reach a cell and more beyond the
// object's box.)
// (exact form; `fcell` = bp_cell_index of the ray origin's cell)
RLG_HD bool ray_proxy_listed(uint16_t hist, int fcell) {
    const int cell = (int)(hist >> 3);
    const int ci = cell / (BP_CELLS_Y * BP_CELLS_Z), cj = (cell / BP_CELLS_Z) % BP_CELLS_Y, ck = cell % BP_CELLS_Z;
    const int fi = fcell / (BP_CELLS_Y * BP_CELLS_Z), fj = (fcell / BP_CELLS_Z) % BP_CELLS_Y, fk = fcell % BP_CELLS_Z;
    const int di = fi - ci, dj = fj - cj, dk = fk - ck;
    return di >= -1 && di <= 1 && dj >= -1 && dj <= 1 && dk >= -1 && dk <= 1;
}
// an object the ray's box does NOT come near, but whose proxy may be on the list of the ray's cell: the cast has to run all the same (a cold call: the common
// path keeps its registers; `fcell - cell` beyond one cell in every direction is rejected inline)
constexpr int RAY_CELL_REACH = BP_CELLS_Y * BP_CELLS_Z + BP_CELLS_Z + 1;
// the cast's first pass through its loop, restated (arena_simplex.h ray_convex_cast: the initial support towards the ray, then one support along v): a ray that points
// away from the object, or reaches it only beyond its end, leaves there -- `return false`, or a lambda > 1 that no closest-hit test accepts -- before the simplex is
// touched.  What nearly every far object ends with; only the others pay for the cast.
RLG_HD bool ray_cast_first_pass_misses(V3 from, V3 to, const M3& Rb, V3 ob, V3 half, float radius) {
    const V3 r = (to - from) - (ob - ob);
    V3 sup_b = cast_support_b(Rb, ob, half, radius, r);
    const V3 v = from - sup_b;
    if (!(len2(v) > 0.0001f)) return false;
    sup_b = cast_support_b(Rb, ob, half, radius, v);
    const V3 w = from - sup_b;
    const float VdotW = dot(v, w);
    if (VdotW > 0.f) {
        const float VdotR = dot(v, r);
        if (VdotR >= -(SIMD_EPS * SIMD_EPS)) return true;
        if (0.f - VdotW / VdotR > 1.0f) return true;
    }
    return false;
}
RLG_HD_COLD void ray_far_proxy(uint16_t hist, int fcell, V3 from, V3 to, const M3& R, V3 center, V3 half, float radius, int kind, RayHit& best) {
    if (!ray_proxy_listed(hist, fcell)) return;
    if (RLG_LIKELY(ray_cast_first_pass_misses(from, to, R, center, half, radius))) return;
    ray_convex_hit(from, to, R, center, half, radius, kind, best);
}
RLG_HD bool ray_far_proxy_possible(uint16_t hist, int fcell) {
    const int d = fcell - (int)(hist >> 3);
    return hist != 0 && d <= RAY_CELL_REACH && d >= -RAY_CELL_REACH;
}
template <int NC>
RLG_HD void ray_ball_and_cars(const Arena<NC>& A, int self_car, V3 from, V3 to, RayHit& best) {
    const bool lists = (A.mut.flags & MUT_RAY_PROXY_LISTS) != 0;   // (off, the default: one flag read is all the far branch costs)
    int fcell = 0;
    if (RLG_UNLIKELY(lists)) { int fi, fj, fk; bp_cell_of(from, fi, fj, fk); fcell = bp_cell_index(fi, fj, fk); }     // GetCellIdx(rayFrom)
    {
        const float r = K::BALL_RADIUS * UU2BT;
        const V3 bp = A.ball.b.pos;
        // the cast runs in the ball's basis (BallState::rotMat), as btCollisionWorld::rayTestSingle's convex cast does
        if (RLG_UNLIKELY(ray_box_near(from, to, bp - v3(r, r, r), bp + v3(r, r, r)))) ray_convex_hit(from, to, A.ball.b.rot, bp, v3(0, 0, 0), r, 1, best);
        else if (RLG_UNLIKELY(lists) && ray_far_proxy_possible(A.bp_hist[0], fcell)) ray_far_proxy(A.bp_hist[0], fcell, from, to, A.ball.b.rot, bp, v3(0, 0, 0), r, 1, best);
    }
    // other cars' hitboxes.  A car that is demoed, or was respawned this tick, has no contact response (Car.cpp:69-80) but its rigid body
    // stays in the world where it stopped: the ray test finds the CLOSEST object first and only then asks whether it responds
    // (btDefaultVehicleRaycaster.cpp:36-51) -- so such a body does not let the ray through to the ground behind it, it turns the whole ray
    // into a miss.  (Found on `3v3_kickoff`, tick 341, with tools/raw_divergence.py: a wheel over a wreck.)
    constexpr int GHOST = 2 + NC;
    for (int k = 0; k < NC; k++) {
        if (k == self_car) continue;
        const Car& o = A.cars[k];
        if (o.flags & CF_ABSENT) continue;      // an empty slot of a one-team env: no body at all
        bool ghost = (o.flags & CF_IS_DEMOED) || o.frozen;
        // ... from ITS OWN pre-tick on: Car::_PreTickUpdate is what clears the body's contact response (Car.cpp:69-80), so on the first
        // tick after a demolition the cars the arena visits BEFORE the wreck still find a body like any other (Arena.cpp:716-812)
        if (ghost && car_demolished_last_tick(o, A.mut.respawn_delay) && car_rank_of(A, self_car) < car_rank_of(A, k)) ghost = false;
        const M3 R = (o.flags & CF_IS_DEMOED) ? car_ghost_rot(o) : o.b.rot;
        const V3 center = o.b.pos + R * hitbox_off();
        const V3 h = hitbox_half();
        const V3 e = abs_rows_dot(R, h);
        if (RLG_LIKELY(!ray_box_near(from, to, center - e, center + e))) {
            if (RLG_UNLIKELY(lists) && ray_far_proxy_possible(A.bp_hist[1 + k], fcell)) ray_far_proxy(A.bp_hist[1 + k], fcell, from, to, R, center, h, 0.f, ghost ? GHOST : 2 + k, best);
            continue;
        }
        ray_convex_hit(from, to, R, center, h, 0.f, ghost ? GHOST : 2 + k, best);
    }
    if (best.kind == GHOST) best.kind = -1;
}

// embree closest point on triangle (SphereTriangleDetector.cpp:87-129). `feature` out: 0 face, 1..3 vertex a/b/c, 4 edge ab, 5 edge ac, 6 edge bc
RLG_HD V3 closest_point_triangle(V3 p, V3 a, V3 b, V3 c, int& feature) {
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { feature = 1; return a; }
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { feature = 2; return b; }
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { feature = 3; return c; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); feature = 4; return a + v * ab; }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { float v = d2 / (d2 - d6); feature = 5; return a + v * ac; }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { float v = (d4 - d3) / ((d4 - d3) + (d5 - d6)); feature = 6; return b + v * (c - b); }
    float denom = 1.f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    feature = 0;
    return a + v * ab + w * ac;
}

// SphereTriangleDetector::pointInTriangle (the reference's replacement, SphereTriangleDetector.cpp:245-...):
// inside iff the point is on the inner side of the three edge planes
RLG_HD bool point_in_triangle(V3 p, V3 v0, V3 v1, V3 v2, V3 n) {
    V3 e1 = v1 - v0, e2 = v2 - v1, e3 = v0 - v2;
    float r1 = dot(cross(e1, n), p - v0), r2 = dot(cross(e2, n), p - v1), r3 = dot(cross(e3, n), p - v2);
    return (r1 > 0 && r2 > 0 && r3 > 0) || (r1 <= 0 && r2 <= 0 && r3 <= 0);
}

// btAdjustInternalEdgeContacts (btInternalEdgeUtility.cpp:413-797, normalAdjustFlags = 0), run by the contact-added callback on every
// new point against the mesh (Arena.cpp:275-279): a contact that lies within 0.1 of the triangle's closest shared edge gets the
// face normal when the edge is flat / concave towards the contact normal, a normal clamped into the edge's Voronoi wedge when it is
// convex; the point on the triangle is re-projected from the (unchanged) point on the body.  The mesh body sits at the identity.
RLG_HD V3 nearest_on_segment(V3 p, V3 l0, V3 l1) {
    V3 d = l1 - l0;
    if (len2(d) < SIMD_EPS * SIMD_EPS) return l0;
    float delta = dot(p - l0, d) / dot(d, d);
    if (delta < 0.f) delta = 0.f; else if (delta > 1.f) delta = 1.f;
    return l0 + d * delta;
}
RLG_HD_T9 void adjust_internal_edge(const MeshTri& t, V3& pb, V3& n, float dist) {
    if (!(t.edge_flags >> 31)) return;
    const V3 pa = pb + n * dist;                      // m_positionWorldOnA (btManifoldResult.cpp:117)
    const V3 v[3] = {v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z)};
    const V3 tri_normal = normalized(cross(v[1] - v[0], v[2] - v[0]));
    const V3 contact = pb;
    const V3 local_n = normalized(n);
    const float TWO_PI = 6.283185307179586232f;
    int best = -1; float best_d = 1e18f;
    for (int e = 0; e < 3; e++) {
        if (!(fabsf(t.edge_angle[e]) < TWO_PI)) continue;
        float l = len(contact - nearest_on_segment(contact, v[e], v[(e + 1) % 3]));
        if (l < best_d) { best = e; best_d = l; }
    }
    if (best < 0 || !(best_d < 0.1f)) return;         // m_edgeDistanceThreshold
    const int e = best;
    const float angle = t.edge_angle[e];
    bool concave = false;
    if (angle == 0.f) concave = true;
    else {
        const V3 edge = v[e] - v[(e + 1) % 3];
        const bool convex = (t.edge_flags >> e) & 1u;
        const float swap = convex ? 1.f : -1.f;
        const V3 nA = swap * tri_normal;
        V3 cnb = quat_rotate(quat_axis_angle(edge, angle), tri_normal);
        if ((t.edge_flags >> (3 + e)) & 1u) cnb *= -1.f;
        const V3 nB = swap * cnb;
        const bool back_facing = (dot(local_n, nA) < 0.f) && (dot(local_n, nB) < 0.f);
        if (back_facing) concave = true;
        else {
            // btClampNormal (:375-410); edges 1 and 2 pass the un-normalised local normal (:607, :676)
            const V3 ln = e == 0 ? local_n : n;
            const V3 edge_cross = normalized(cross(edge, nA));
            const float cur = rl_atan2f(dot(ln, edge_cross), dot(ln, nA));
            if ((angle < 0.f && cur < angle) || (angle >= 0.f && cur > angle)) {
                const V3 clamped = quat_to_m3(quat_axis_angle(edge, angle - cur)) * ln;
                if (dot(clamped, tri_normal) > 0.f) { n = clamped; pb = pa - n * dist; }
            }
        }
    }
    if (concave) {
        if (dot(tri_normal, local_n) < 0.f) return;
        n = tri_normal; pb = pa - n * dist;
    }
}

// sphere (ball) vs one triangle: SphereTriangleDetector::collide (:139-241) + the internal-edge snap
RLG_HD_T5 bool sphere_triangle(V3 c, float radius, float thresh, const MeshTri& t, V3& point, V3& normal, float& depth) {
    V3 v0 = v3(t.v0x, t.v0y, t.v0z), v1 = v3(t.v1x, t.v1y, t.v1z), v2 = v3(t.v2x, t.v2y, t.v2z);
    float rwt = radius + thresh;
    V3 n = cross(v1 - v0, v2 - v0);
    float l2 = len2(n);
    if (l2 < SIMD_EPS * SIMD_EPS) return false;
    n = vdiv_bt(n, sqrtf(l2));
    float dplane = dot(c - v0, n);
    bool back_side = false;
    if (dplane < 0.f) { dplane *= -1.f; n = n * -1.f; back_side = true; }
    if (!(dplane < rwt)) return false;
    V3 cp; int feature = 0; bool has = false;
    if (point_in_triangle(c, v0, v1, v2, n)) { has = true; cp = c - n * dplane; }
    else {
        V3 q = closest_point_triangle(c, v0, v1, v2, feature);
        float ds = len2(q - c);
        if (ds < rwt * rwt) { has = true; cp = q; }
    }
    if (!has) return false;
    V3 ctc = c - cp;
    float ds = len2(ctc);
    if (!(ds < rwt * rwt)) return false;
    if (ds > SIMD_EPS) {
        float d = sqrtf(ds);
        normal = normalized(ctc);   // resultNormal.normalize() (SphereTriangleDetector.cpp:227-228): the SSE normalise, not a division by d
        point = cp; depth = -(radius - d);
    } else { normal = n; point = cp; depth = -radius; }
    return true;
}

// Convex polygon of <= 8 points for the Sutherland-Hodgman clips below.  Every loop over it is fully unrolled with compile-time
// slot numbers (reads select on `i < n`, writes go through a compare chain on the insert position), so on the device the whole
// polygon lives in registers: as V3 arrays indexed by loop variables it sat in scratch memory, and the clip was a chain of
// dependent scratch round trips -- the slowest thing in a contact-heavy tick.
struct Poly8 {
    V3 p[8];
    int n;
};
RLG_HD void poly_push(Poly8& P, V3 v) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 8; s++) if (P.n == s) P.p[s] = v;
    if (P.n < 8) P.n++;
}
// clip against the half space dot(nrm,p) <= off
RLG_HD void clip_poly(Poly8& P, V3 nrm, float off) {
    const int n = P.n;
    {   // most planes of a contact do not cut at all: with every vertex on the inner side the loop below copies P vertex by vertex,
        // with every vertex outside it leaves nothing -- both without the per-vertex insert chains
        bool all_in = true, all_out = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 8; i++)
            if (i < n) { float da = dot(nrm, P.p[i]) - off; all_in = all_in && (da <= 0.f); all_out = all_out && (da > 0.f); }
        if (all_in) return;
        if (all_out) { P.n = 0; return; }
    }
    Poly8 O; O.n = 0;
    const V3 first = P.p[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 8; i++) {
        if (i < n) {
            V3 a = P.p[i], b = (i + 1 < n) ? P.p[(i + 1) & 7] : first;
            float da = dot(nrm, a) - off, db = dot(nrm, b) - off;
            if (da <= 0.f) poly_push(O, a);
            if ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f)) {
                float t = da / (da - db);
                poly_push(O, a + (b - a) * t);
            }
        }
    }
    P = O;
}

// box (center bc, basis R, half extents h) vs triangle, SAT + clipping. Emits candidates with the normal
// pointing from the triangle towards the box.
template <int CAP>
RLG_HD_NOINLINE void box_triangle(V3 bc, const M3& R, V3 h, const MeshTri& t, float thresh, Cand (&cs)[CAP], int& nc) {
    V3 p[3] = {tmul(R, v3(t.v0x, t.v0y, t.v0z) - bc), tmul(R, v3(t.v1x, t.v1y, t.v1z) - bc), tmul(R, v3(t.v2x, t.v2y, t.v2z) - bc)};
    V3 e[3] = {p[1] - p[0], p[2] - p[1], p[0] - p[2]};
    V3 n = cross(e[0], p[2] - p[0]);
    float nl = len(n);
    if (nl < 1e-12f) return;
    n = vdiv_bt(n, nl);
    float d = dot(n, p[0]);
    if (d > 0.f) { n = -n; d = -d; }  // now the box centre is on the +n side: signed distance of centre = -d >= 0
    float rn = h.x * fabsf(n.x) + h.y * fabsf(n.y) + h.z * fabsf(n.z);
    float best_sep = (-d) - rn; int best_type = 0, best_i = 0, best_j = 0; V3 best_axis = n;
    if (best_sep > thresh) return;
    // box face axes
    for (int i = 0; i < 3; i++) {
        float mn = fminf(get(p[0], i), fminf(get(p[1], i), get(p[2], i)));
        float mx = fmaxf(get(p[0], i), fmaxf(get(p[1], i), get(p[2], i)));
        float hi_ = get(h, i);
        float s1 = mn - hi_, s2 = -mx - hi_;
        float s = fmaxf(s1, s2);
        if (s > thresh) return;
        if (s > best_sep + 1e-5f) { best_sep = s; best_type = 1; best_i = i; best_j = (s1 > s2) ? 1 : -1; }
    }
    // edge x edge axes
    for (int i = 0; i < 3; i++) {
        V3 ei = v3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f);
        for (int j = 0; j < 3; j++) {
            V3 a = cross(ei, e[j]);
            float al = len(a);
            if (al < 1e-6f) continue;
            a = vdiv_bt(a, al);
            float t0 = dot(a, p[0]), t1 = dot(a, p[1]), t2 = dot(a, p[2]);
            float mn = fminf(t0, fminf(t1, t2)), mx = fmaxf(t0, fmaxf(t1, t2));
            float r = h.x * fabsf(a.x) + h.y * fabsf(a.y) + h.z * fabsf(a.z);
            float s1 = mn - r, s2 = -mx - r;
            float s = fmaxf(s1, s2);
            if (s > thresh) return;
            if (s > best_sep + 1e-3f) {  // prefer face axes on near ties (as dBoxBox2's fudge factor does)
                best_sep = s; best_type = 2; best_i = i; best_j = j;
                best_axis = (s1 > s2) ? -a : a;  // points from triangle towards the box centre
            }
        }
    }
    if (best_type == 0) {
        // reference = triangle plane; incident = box face most anti-parallel to n
        int k = 0; float mk = fabsf(n.x);
        if (fabsf(n.y) > mk) { k = 1; mk = fabsf(n.y); }
        if (fabsf(n.z) > mk) { k = 2; }
        float sg = get(n, k) > 0.f ? -1.f : 1.f;
        int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        Poly8 quad; quad.n = 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            float s1 = (q == 0 || q == 3) ? -1.f : 1.f, s2 = (q < 2) ? -1.f : 1.f;
            float c0 = sg * get(h, k), c1 = s1 * get(h, k1), c2 = s2 * get(h, k2);   // components along axes k, k1, k2
            quad.p[q] = v3(k == 0 ? c0 : (k1 == 0 ? c1 : c2), k == 1 ? c0 : (k1 == 1 ? c1 : c2), k == 2 ? c0 : (k1 == 2 ? c1 : c2));
        }
        // clip against the triangle's edge planes (inward side), computed with the un-flipped winding
        V3 tn = cross(e[0], p[2] - p[0]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 3; j++) {
            if (quad.n <= 0) continue;
            V3 en = cross(e[j], tn);  // outward
            float enl = len(en);
            if (enl < 1e-12f) continue;
            en = vdiv_bt(en, enl);
            clip_poly(quad, en, dot(en, p[j]));
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 8; q++) {
            if (q >= quad.n) continue;
            float dist = dot(n, quad.p[q]) - d;
            if (dist < thresh) {
                Cand c; c.n = R * n; c.dist = dist; c.pb = bc + R * (quad.p[q] - n * dist);
                cand_add(cs, nc, c);
            }
        }
    } else if (best_type == 1) {
        // reference = box face (axis best_i, side best_j), incident = triangle clipped to the face rectangle
        int k = best_i, k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        float side = (float)best_j;  // +1: triangle is on the +axis side
        Poly8 poly; poly.n = 3;
        poly.p[0] = p[0]; poly.p[1] = p[1]; poly.p[2] = p[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            if (poly.n <= 0) continue;
            int ax = (q < 2) ? k1 : k2; float sg = (q % 2) ? 1.f : -1.f;
            V3 cn = v3(ax == 0 ? sg : 0.f, ax == 1 ? sg : 0.f, ax == 2 ? sg : 0.f);
            clip_poly(poly, cn, get(h, ax));
        }
        V3 fn = v3(k == 0 ? side : 0.f, k == 1 ? side : 0.f, k == 2 ? side : 0.f);  // outward face normal
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 8; q++) {
            if (q >= poly.n) continue;
            float dist = side * get(poly.p[q], k) - get(h, k);
            if (dist < thresh) {
                Cand c; c.n = R * (-fn); c.dist = dist; c.pb = bc + R * poly.p[q];
                cand_add(cs, nc, c);
            }
        }
    } else {
        // edge-edge: box edge along axis best_i at the support corner towards -axis, triangle edge best_j
        V3 a = best_axis;
        V3 corner = v3(a.x > 0 ? -h.x : h.x, a.y > 0 ? -h.y : h.y, a.z > 0 ? -h.z : h.z);
        int i = best_i;
        V3 ei = v3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f);
        V3 pa = corner; if (i == 0) pa.x = 0; else if (i == 1) pa.y = 0; else pa.z = 0;  // midpoint of the box edge
        V3 qb = p[best_j], ub = e[best_j];
        // closest points between line pa + s*ei and segment qb + t*ub
        V3 w = pa - qb;
        float uaub = dot(ei, ub), q1 = dot(ei, w), q2 = -dot(ub, w);
        float ubub = dot(ub, ub);
        float dd = ubub - uaub * uaub;
        float tt = (dd > 1e-12f) ? clampf((uaub * q1 * -1.f + q2 * -1.f) / -dd, 0.f, 1.f) : 0.5f;
        // robust: recompute t directly as the projection of the closest approach
        {
            float den = ubub - uaub * uaub;
            if (den > 1e-12f) tt = clampf((dot(ub, w) - uaub * dot(ei, w)) / den, 0.f, 1.f);
        }
        V3 pt = qb + ub * tt;
        Cand c; c.n = R * a; c.dist = best_sep; c.pb = bc + R * pt;
        cand_add(cs, nc, c);
    }
}

// box vs box (SAT, 15 axes, face clipping / edge-edge). A = (ca,Ra), B = (cb,Rb); normals point from B to A.
template <int CAP>
RLG_HD_NOINLINE void box_box(V3 ca, const M3& Ra, V3 cb, const M3& Rb, V3 h, Cand (&cs)[CAP], int& nc) {
    V3 pp = tmul(Ra, cb - ca);            // B centre in A's frame
    M3 Rr = transpose(Ra) * Rb;           // B axes in A's frame (columns)
    V3 bcol[3] = {col0(Rr), col1(Rr), col2(Rr)};
    float best = -1e30f; int type = -1, bi = 0, bj = 0; V3 axis = v3(1, 0, 0); float flip = 1.f;
    // A's face axes
    for (int i = 0; i < 3; i++) {
        float rb = h.x * fabsf(get(bcol[0], i)) + h.y * fabsf(get(bcol[1], i)) + h.z * fabsf(get(bcol[2], i));
        float s = fabsf(get(pp, i)) - (get(h, i) + rb);
        if (s > 0.f) return;
        if (s > best) { best = s; type = 0; bi = i; flip = get(pp, i) < 0 ? -1.f : 1.f; }
    }
    // B's face axes
    for (int j = 0; j < 3; j++) {
        float ra = h.x * fabsf(bcol[j].x) + h.y * fabsf(bcol[j].y) + h.z * fabsf(bcol[j].z);
        float pj = dot(pp, bcol[j]);
        float s = fabsf(pj) - (ra + get(h, j));
        if (s > 0.f) return;
        if (s > best) { best = s; type = 1; bj = j; flip = pj < 0 ? -1.f : 1.f; }
    }
    // edge-edge
    for (int i = 0; i < 3; i++) {
        V3 ei = v3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f);
        for (int j = 0; j < 3; j++) {
            V3 a = cross(ei, bcol[j]);
            float al = len(a);
            if (al < 1e-5f) continue;
            a = vdiv_bt(a, al);
            float ra = h.x * fabsf(a.x) + h.y * fabsf(a.y) + h.z * fabsf(a.z);
            float rb = h.x * fabsf(dot(a, bcol[0])) + h.y * fabsf(dot(a, bcol[1])) + h.z * fabsf(dot(a, bcol[2]));
            float pa = dot(pp, a);
            float s = fabsf(pa) - (ra + rb);
            if (s > 0.f) return;
            if (s * 1.05f > best + 1e-4f && s > best * 1.05f) {  // fudge: favour face axes (btBoxBoxDetector.cpp fudge_factor)
                best = s; type = 2; bi = i; bj = j; axis = pa < 0 ? -a : a;
            }
        }
    }
    if (type == 0 || type == 1) {
        // reference box = the one owning the axis; incident face from the other box
        bool ref_is_a = (type == 0);
        V3 nrm_a;  // separating direction from A to B in A's frame
        if (ref_is_a) nrm_a = v3(bi == 0 ? flip : 0.f, bi == 1 ? flip : 0.f, bi == 2 ? flip : 0.f);
        else nrm_a = bcol[bj] * flip;
        // express everything in the reference box frame
        M3 Rref = ref_is_a ? m3_identity() : Rr;              // ref axes in A frame (columns)
        V3 cref = ref_is_a ? v3(0, 0, 0) : pp;
        M3 Rinc = ref_is_a ? Rr : m3_identity();
        V3 cinc = ref_is_a ? pp : v3(0, 0, 0);
        V3 nref = ref_is_a ? nrm_a : -nrm_a;                   // outward from ref towards inc, in A frame
        // incident face: face of inc most anti-parallel to nref
        V3 icol[3] = {col0(Rinc), col1(Rinc), col2(Rinc)};
        int k = 0; float mk = fabsf(dot(nref, icol[0]));
        for (int q = 1; q < 3; q++) { float v = fabsf(dot(nref, icol[q])); if (v > mk) { mk = v; k = q; } }
        float sg = dot(nref, icol[k]) > 0.f ? -1.f : 1.f;
        int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        Poly8 quad; quad.n = 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            float s1 = (q == 0 || q == 3) ? -1.f : 1.f, s2 = (q < 2) ? -1.f : 1.f;
            quad.p[q] = cinc + icol[k] * (sg * get(h, k)) + icol[k1] * (s1 * get(h, k1)) + icol[k2] * (s2 * get(h, k2));
        }
        V3 rcol[3] = {col0(Rref), col1(Rref), col2(Rref)};
        int rk = ref_is_a ? bi : bj;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            if (quad.n <= 0) continue;
            int ax = (q < 2) ? (rk + 1) % 3 : (rk + 2) % 3; float s = (q % 2) ? 1.f : -1.f;
            V3 cn = rcol[ax] * s;
            clip_poly(quad, cn, dot(cn, cref) + get(h, ax));
        }
        float face_off = dot(nref, cref) + get(h, rk);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 8; q++) {
            if (q >= quad.n) continue;
            float dist = dot(nref, quad.p[q]) - face_off;  // < 0 when the incident point is inside the reference box
            if (dist < 0.f) {
                Cand c; c.dist = dist;
                // normal from B to A (world). nrm_a points A->B.
                c.n = Ra * (-nrm_a);
                // point on B: if the reference is A the incident point is on B; else project onto B's face
                V3 pb_a = ref_is_a ? quad.p[q] : (quad.p[q] - nref * dist);
                c.pb = ca + Ra * pb_a;
                cand_add(cs, nc, c);
            }
        }
    } else if (type == 2) {
        // edge-edge: support edges
        V3 a = axis;  // A -> B in A frame
        V3 pa = v3(a.x > 0 ? h.x : -h.x, a.y > 0 ? h.y : -h.y, a.z > 0 ? h.z : -h.z);
        if (bi == 0) pa.x = 0; else if (bi == 1) pa.y = 0; else pa.z = 0;
        V3 pb = pp;
        for (int q = 0; q < 3; q++) {
            if (q == bj) continue;
            float s = dot(a, bcol[q]) > 0 ? -1.f : 1.f;
            pb += bcol[q] * (s * get(h, q));
        }
        V3 ua = v3(bi == 0 ? 1.f : 0.f, bi == 1 ? 1.f : 0.f, bi == 2 ? 1.f : 0.f), ub = bcol[bj];
        V3 w = pb - pa;
        float uaub = dot(ua, ub), q1 = dot(ua, w), q2 = -dot(ub, w);
        float dd = 1.f - uaub * uaub;
        float beta = (dd > 1e-6f) ? (uaub * q1 + q2) / dd : 0.f;
        V3 ptb = pb + ub * beta;
        Cand c; c.dist = best; c.n = Ra * (-a); c.pb = ca + Ra * ptb;
        cand_add(cs, nc, c);
    }
}

// ---- box vs box as the reference does it: btBoxBoxDetector = ODE's dBoxBox2 (BulletCollision/CollisionDispatch/btBoxBoxDetector.cpp:
// 264-716), on the two hitboxes WITH margin (:758-765), up to 4 points per call.  Restated with column vectors: u[i] / v[j] are the
// axes of box 1 / box 2.  Output in btManifoldResult::addContactPoint's terms: normal on box 2 pointing at box 1, point on box 2.
// intersectRectQuad2 (:119-183): the rectangle +-h clipped against the quadrilateral p; at most 8 points
RLG_HD int ode_rect_quad(const float h[2], const float p[8], float ret[16]) {
    float qa[16], ra[16];
    for (int i = 0; i < 8; i++) qa[i] = p[i];
    float* q = qa; float* r = ra;
    int nq = 4, nr = 0;
    bool done = false;
    for (int dir = 0; dir <= 1 && !done; dir++) {
        for (int sign = -1; sign <= 1 && !done; sign += 2) {
            float* pq = q; float* pr = r;
            nr = 0;
            for (int i = nq; i > 0; i--) {
                if (sign * pq[dir] < h[dir]) {
                    pr[0] = pq[0]; pr[1] = pq[1]; pr += 2; nr++;
                    if (nr & 8) { q = r; done = true; break; }
                }
                float* nextq = (i > 1) ? pq + 2 : q;
                if ((sign * pq[dir] < h[dir]) ^ (sign * nextq[dir] < h[dir])) {
                    pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (sign * h[dir] - pq[dir]);
                    pr[dir] = sign * h[dir];
                    pr += 2; nr++;
                    if (nr & 8) { q = r; done = true; break; }
                }
                pq += 2;
            }
            if (done) break;
            q = r;
            r = (q == ra) ? qa : ra;
            nq = nr;
        }
    }
    for (int i = 0; i < nr * 2; i++) ret[i] = q[i];
    return nr;
}
// cullPoints2 (:194-262): m of the n points, spread by angle around the centroid, starting with i0
RLG_HD void ode_cull_points(int n, const float p[], int m, int i0, int iret[]) {
    const float M_PI_ODE = 3.14159265f;
    float a, cx, cy, q;
    if (n == 1) { cx = p[0]; cy = p[1]; }
    else if (n == 2) { cx = 0.5f * (p[0] + p[2]); cy = 0.5f * (p[1] + p[3]); }
    else {
        a = 0; cx = 0; cy = 0;
        for (int i = 0; i < (n - 1); i++) {
            q = p[i * 2] * p[i * 2 + 3] - p[i * 2 + 2] * p[i * 2 + 1];
            a += q; cx += q * (p[i * 2] + p[i * 2 + 2]); cy += q * (p[i * 2 + 1] + p[i * 2 + 3]);
        }
        q = p[n * 2 - 2] * p[1] - p[0] * p[n * 2 - 1];
        if (fabsf(a + q) > SIMD_EPS) a = 1.f / (3.0f * (a + q)); else a = 1e18f;
        cx = a * (cx + q * (p[n * 2 - 2] + p[0]));
        cy = a * (cy + q * (p[n * 2 - 1] + p[1]));
    }
    float A[8]; int avail[8];
    for (int i = 0; i < n; i++) { A[i] = rl_atan2f(p[i * 2 + 1] - cy, p[i * 2] - cx); avail[i] = 1; }
    avail[i0] = 0;
    iret[0] = i0;
    for (int j = 1; j < m; j++) {
        a = (float)j * (2 * M_PI_ODE / m) + A[i0];
        if (a > M_PI_ODE) a -= 2 * M_PI_ODE;
        float maxdiff = 1e9f, diff;
        iret[j] = i0;
        for (int i = 0; i < n; i++) {
            if (avail[i]) {
                diff = fabsf(A[i] - a);
                if (diff > M_PI_ODE) diff = 2 * M_PI_ODE - diff;
                if (diff < maxdiff) { maxdiff = diff; iret[j] = i; }
            }
        }
        avail[iret[j]] = 0;
    }
}
RLG_HD_NOINLINE void box_box_ode(V3 p1, const M3& R1, V3 p2, const M3& R2, V3 h, Cand (&cs)[4], int& nc) {
    const float fudge_factor = 1.05f;
    const V3 u[3] = {col0(R1), col1(R1), col2(R1)}, v[3] = {col0(R2), col1(R2), col2(R2)};
    const V3 p = p2 - p1;
    const float pp[3] = {dot(u[0], p), dot(u[1], p), dot(u[2], p)};
    const float A[3] = {(2.f * h.x) * 0.5f, (2.f * h.y) * 0.5f, (2.f * h.z) * 0.5f};
    const float* B = A;
    float R[3][3], Q[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = dot(u[i], v[j]); Q[i][j] = fabsf(R[i][j]); }
    float s = -3.402823466e+38f, s2, l;
    int invert = 0, code = 0;
    V3 normal_r = v3(0, 0, 0), normal_c = v3(0, 0, 0); bool use_r = false;
#define RLG_TST(expr1, expr2, nrm, cc) { const float e1_ = (expr1); s2 = fabsf(e1_) - (expr2); if (s2 > 0) return; if (s2 > s) { s = s2; normal_r = (nrm); use_r = true; invert = (e1_ < 0); code = (cc); } }
    RLG_TST(pp[0], (A[0] + B[0] * Q[0][0] + B[1] * Q[0][1] + B[2] * Q[0][2]), u[0], 1);
    RLG_TST(pp[1], (A[1] + B[0] * Q[1][0] + B[1] * Q[1][1] + B[2] * Q[1][2]), u[1], 2);
    RLG_TST(pp[2], (A[2] + B[0] * Q[2][0] + B[1] * Q[2][1] + B[2] * Q[2][2]), u[2], 3);
    RLG_TST(dot(v[0], p), (A[0] * Q[0][0] + A[1] * Q[1][0] + A[2] * Q[2][0] + B[0]), v[0], 4);
    RLG_TST(dot(v[1], p), (A[0] * Q[0][1] + A[1] * Q[1][1] + A[2] * Q[2][1] + B[1]), v[1], 5);
    RLG_TST(dot(v[2], p), (A[0] * Q[0][2] + A[1] * Q[1][2] + A[2] * Q[2][2] + B[2]), v[2], 6);
#undef RLG_TST
#define RLG_TST(expr1, expr2, n1, n2, n3, cc) { const float e1_ = (expr1); s2 = fabsf(e1_) - (expr2); if (s2 > SIMD_EPS) return; l = sqrtf((n1) * (n1) + (n2) * (n2) + (n3) * (n3)); \
        if (l > SIMD_EPS) { s2 /= l; if (s2 * fudge_factor > s) { s = s2; use_r = false; normal_c = v3((n1) / l, (n2) / l, (n3) / l); invert = (e1_ < 0); code = (cc); } } }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] += 1.0e-5f;
    RLG_TST(pp[2] * R[1][0] - pp[1] * R[2][0], (A[1] * Q[2][0] + A[2] * Q[1][0] + B[1] * Q[0][2] + B[2] * Q[0][1]), 0.f, -R[2][0], R[1][0], 7);
    RLG_TST(pp[2] * R[1][1] - pp[1] * R[2][1], (A[1] * Q[2][1] + A[2] * Q[1][1] + B[0] * Q[0][2] + B[2] * Q[0][0]), 0.f, -R[2][1], R[1][1], 8);
    RLG_TST(pp[2] * R[1][2] - pp[1] * R[2][2], (A[1] * Q[2][2] + A[2] * Q[1][2] + B[0] * Q[0][1] + B[1] * Q[0][0]), 0.f, -R[2][2], R[1][2], 9);
    RLG_TST(pp[0] * R[2][0] - pp[2] * R[0][0], (A[0] * Q[2][0] + A[2] * Q[0][0] + B[1] * Q[1][2] + B[2] * Q[1][1]), R[2][0], 0.f, -R[0][0], 10);
    RLG_TST(pp[0] * R[2][1] - pp[2] * R[0][1], (A[0] * Q[2][1] + A[2] * Q[0][1] + B[0] * Q[1][2] + B[2] * Q[1][0]), R[2][1], 0.f, -R[0][1], 11);
    RLG_TST(pp[0] * R[2][2] - pp[2] * R[0][2], (A[0] * Q[2][2] + A[2] * Q[0][2] + B[0] * Q[1][1] + B[1] * Q[1][0]), R[2][2], 0.f, -R[0][2], 12);
    RLG_TST(pp[1] * R[0][0] - pp[0] * R[1][0], (A[0] * Q[1][0] + A[1] * Q[0][0] + B[1] * Q[2][2] + B[2] * Q[2][1]), -R[1][0], R[0][0], 0.f, 13);
    RLG_TST(pp[1] * R[0][1] - pp[0] * R[1][1], (A[0] * Q[1][1] + A[1] * Q[0][1] + B[0] * Q[2][2] + B[2] * Q[2][0]), -R[1][1], R[0][1], 0.f, 14);
    RLG_TST(pp[1] * R[0][2] - pp[0] * R[1][2], (A[0] * Q[1][2] + A[1] * Q[0][2] + B[0] * Q[2][1] + B[1] * Q[2][0]), -R[1][2], R[0][2], 0.f, 15);
#undef RLG_TST
    if (!code) return;
    V3 normal = use_r ? normal_r : (R1 * normal_c);
    if (invert) normal = -normal;
    const float depth = -s;
    auto emit = [&](V3 pt, float dep) { if (nc < 4) { Cand c; c.n = -normal; c.pb = pt; c.dist = -dep; cs[nc++] = c; } };
    if (code > 6) {
        V3 pa = p1;
        for (int j = 0; j < 3; j++) { float sign = (dot(normal, u[j]) > 0) ? 1.0f : -1.0f; pa = pa + u[j] * (sign * A[j]); }
        V3 pb = p2;
        for (int j = 0; j < 3; j++) { float sign = (dot(normal, v[j]) > 0) ? -1.0f : 1.0f; pb = pb + v[j] * (sign * B[j]); }
        const V3 ua = u[(code - 7) / 3], ub = v[(code - 7) % 3];
        // dLineClosestApproach (:85-110)
        V3 d = pb - pa;
        float uaub = dot(ua, ub), q1 = dot(ua, d), q2 = -dot(ub, d), dd = 1 - uaub * uaub, beta;
        if (dd <= 0.0001f) beta = 0; else { dd = 1.f / dd; beta = (uaub * q1 + q2) * dd; }
        pb = pb + ub * beta;
        emit(pb, depth);
        return;
    }
    const bool first = code <= 3;
    const V3* Ra = first ? u : v; const V3* Rb = first ? v : u;
    const V3 pa = first ? p1 : p2, pb = first ? p2 : p1;
    const float* Sa = A; const float* Sb = A;
    const V3 normal2 = first ? normal : -normal;
    const float nr[3] = {dot(Rb[0], normal2), dot(Rb[1], normal2), dot(Rb[2], normal2)};
    const float anr[3] = {fabsf(nr[0]), fabsf(nr[1]), fabsf(nr[2])};
    int lanr, a1, a2;
    if (anr[1] > anr[0]) { if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
    else { if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
    V3 center;
    if (nr[lanr] < 0) center = (pb - pa) + Rb[lanr] * Sb[lanr]; else center = (pb - pa) - Rb[lanr] * Sb[lanr];
    const int codeN = first ? code - 1 : code - 4;
    int code1, code2;
    if (codeN == 0) { code1 = 1; code2 = 2; } else if (codeN == 1) { code1 = 0; code2 = 2; } else { code1 = 0; code2 = 1; }
    float quad[8];
    const float c1 = dot(center, Ra[code1]), c2 = dot(center, Ra[code2]);
    float m11 = dot(Ra[code1], Rb[a1]), m12 = dot(Ra[code1], Rb[a2]), m21 = dot(Ra[code2], Rb[a1]), m22 = dot(Ra[code2], Rb[a2]);
    {
        const float k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
        quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4; quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
        quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4; quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
    }
    const float rect[2] = {Sa[code1], Sa[code2]};
    float ret[16];
    const int n = ode_rect_quad(rect, quad, ret);
    if (n < 1) return;
    V3 point[8]; float dep[8];
    const float det1 = 1.f / (m11 * m22 - m12 * m21);
    m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
    int cnum = 0;
    for (int j = 0; j < n; j++) {
        const float k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2);
        const float k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
        point[cnum] = center + Rb[a1] * k1 + Rb[a2] * k2;
        // (component-wise in the reference: center[i] + k1 * Rb[i][a1] + k2 * Rb[i][a2])
        dep[cnum] = Sa[codeN] - dot(normal2, point[cnum]);
        if (dep[cnum] >= 0) { ret[cnum * 2] = ret[j * 2]; ret[cnum * 2 + 1] = ret[j * 2 + 1]; cnum++; }
    }
    if (cnum < 1) return;
    int maxc = 4;
    if (maxc > cnum) maxc = cnum;
    if (cnum <= maxc) {
        for (int j = 0; j < cnum; j++) {
            if (first) emit(point[j] + pa, dep[j]);
            else emit((point[j] + pa) - normal * dep[j], dep[j]);
        }
    } else {
        int i1 = 0; float maxdepth = dep[0];
        for (int i = 1; i < cnum; i++) if (dep[i] > maxdepth) { maxdepth = dep[i]; i1 = i; }
        int iret[8];
        ode_cull_points(cnum, ret, maxc, i1, iret);
        for (int j = 0; j < maxc; j++) {
            V3 pos = point[iret[j]] + pa;
            if (first) emit(pos, dep[iret[j]]);
            else emit(pos - normal * dep[iret[j]], dep[iret[j]]);
        }
    }
}

// sphere vs rounded box (core = half extents - margin, GJK margins: btConvexConvexAlgorithm / btGjkPairDetector)
RLG_HD_NOINLINE bool sphere_box(V3 sc, float radius, V3 bc, const M3& R, V3 h, float thresh, V3& pb, V3& n, float& dist) {
    V3 hc = v3(h.x - BOX_MARGIN, h.y - BOX_MARGIN, h.z - BOX_MARGIN);
    V3 l = tmul(R, sc - bc);
    V3 q = v3(clampf(l.x, -hc.x, hc.x), clampf(l.y, -hc.y, hc.y), clampf(l.z, -hc.z, hc.z));
    V3 d = l - q;
    float dl2 = len2(d);
    V3 nl; float core_dist;
    if (dl2 > SIMD_EPS * SIMD_EPS) {
        core_dist = sqrtf(dl2);
        nl = vdiv_bt(d, core_dist);
    } else {
        float px = hc.x - fabsf(l.x), py = hc.y - fabsf(l.y), pz = hc.z - fabsf(l.z);
        if (px <= py && px <= pz) { nl = v3(l.x < 0 ? -1.f : 1.f, 0, 0); core_dist = -px; q.x = nl.x * hc.x; }
        else if (py <= pz) { nl = v3(0, l.y < 0 ? -1.f : 1.f, 0); core_dist = -py; q.y = nl.y * hc.y; }
        else { nl = v3(0, 0, l.z < 0 ? -1.f : 1.f); core_dist = -pz; q.z = nl.z * hc.z; }
    }
    dist = core_dist - (BOX_MARGIN + radius);
    if (!(dist < thresh)) return false;
    n = R * nl;
    pb = bc + R * (q + nl * BOX_MARGIN);
    return true;
}

}  // namespace rlg
