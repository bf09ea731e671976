// arena_gjk.h — closest points between the car's hitbox and one mesh triangle, as the reference computes them.
//
// The reference sends every (box, triangle) pair through btConvexConvexAlgorithm -> btGjkPairDetector with btVoronoiSimplexSolver
// (BulletCollision/CollisionDispatch/btConvexConvexAlgorithm.cpp:270-330, NarrowPhaseCollision/btGjkPairDetector.cpp:686-959,
// NarrowPhaseCollision/btVoronoiSimplexSolver.cpp).  GJK works on the CORE shapes -- the box shrunk by its collision margin, the bare
// triangle -- and adds the margins afterwards, so the hitbox it collides is a ROUNDED box: a corner that a sharp box would already
// have within the contact threshold is farther away, and the one contact point a pair yields is GJK's closest point, which for
// parallel features (roof flat on a wall) is whatever vertex combination the iteration ends on.  Both facts decide which contacts
// exist, so the iteration is restated here step by step (same start direction, same support tie-breaks, same simplex reduction,
// same termination tests, fp32) instead of being replaced by an analytic box-triangle distance.
//
// Penetration deeper than the margin (cores overlap, or a degenerate ending with the cores closer than 0.01): the reference then asks
// btGjkEpaPenetrationDepthSolver -- a second GJK and EPA on the margin-inflated shapes -- restated in arena_epa.h and consumed below
// exactly as btGjkPairDetector.cpp:847-927 consumes it.
#pragma once
#include "arena_world.h"
#include "arena_simplex.h"
#include "arena_epa.h"

// Where the penetration-depth solver keeps its state (arena_epa.h).  Host default: a full-size arena on the stack.  The device kernels
// define these before including this header (rlgpu_env.hip): a small arena in LDS shared by the wavefront's lanes one at a time, and a
// full-size one in global memory for the queries that do not fit.
#if !defined(RLG_EPA_ARENA_DECL) && defined(RLG_EPA_HOST_TWO_ARENAS)   // host test build: the device's two-arena scheme (small first, full-size on overflow)
#define RLG_EPA_ARENA_DECL alignas(16) unsigned char epa_mem_[epa_arena_bytes(EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES)]; alignas(16) unsigned char epa_mem_s_[epa_arena_bytes(14, 34)]; \
    EpaArena epa_small_ = epa_arena_at(epa_mem_s_, 14, 34); EpaArena epa_bigv_ = epa_arena_at(epa_mem_, EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); EpaArena* epa_big_ = &epa_bigv_;
#define RLG_EPA_SERIALIZE_BEGIN
#define RLG_EPA_SERIALIZE_END
#define RLG_EPA_BIG_PASS(rc_, CALL)
#define RLG_EPA_COUNT_BIG() RLG_EPA_BIG_HOOK()
#endif
#ifndef RLG_EPA_ARENA_DECL
#define RLG_EPA_ARENA_DECL alignas(16) unsigned char epa_mem_[epa_arena_bytes(EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES)]; \
    EpaArena epa_small_ = epa_arena_at(epa_mem_, EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); EpaArena* epa_big_ = nullptr;
#define RLG_EPA_SERIALIZE_BEGIN
#define RLG_EPA_SERIALIZE_END
#define RLG_EPA_BIG_PASS(rc_, CALL)
#define RLG_EPA_COUNT_BIG() ((void)0)
#endif

namespace rlg {

struct GjkOut { V3 n, pb; float dist; };
constexpr float GJK_REL_ERROR2 = 1.0e-6f;

// The penetration case of btGjkPairDetector (btGjkPairDetector.cpp:847-927) on what the first GJK left in `p`: out of line, so that the
// rare deep contact does not weigh on the register allocation of the GJK loop every pair runs.  false: the arena was too small and no
// full-size one is available.
struct GjkPen { bool valid; float distance; V3 pa, pb, normal; };
RLG_HD_NOINLINE bool gjk_penetration(const M3& R, V3 oa, V3 core, float margin_a, V3 ob, V3 t0, V3 t1, V3 t2, float margin_b, GjkPen& p) {
    RLG_EPA_ARENA_DECL
    const float margin = margin_a + margin_b;
    const EpaShapes es = epa_shapes(R, oa, core, margin_a, ob, t0, t1, t2, margin_b);
    PenDepth pd; pd.v = pd.wa = pd.wb = v3(0, 0, 0);
    int rc = EPA_ARENA_FULL;
    RLG_EPA_SERIALIZE_BEGIN
    rc = epa_calc_pen_depth(epa_small_, es, pd);
    if (rc == EPA_ARENA_FULL && epa_big_) { RLG_EPA_COUNT_BIG(); rc = epa_calc_pen_depth(*epa_big_, es, pd); }
    RLG_EPA_SERIALIZE_END
    RLG_EPA_BIG_PASS(rc, rc = epa_calc_pen_depth(*epa_big_, es, pd))
    if (rc == EPA_ARENA_FULL) return false;
    const V3 axis = pd.v;                                       // m_cachedSeparatingAxis
    if (rc == 1) {
        V3 tn = pd.wb - pd.wa;
        float l2 = len2(tn);
        if (l2 <= SIMD_EPS * SIMD_EPS) { tn = axis; l2 = len2(axis); }
        if (l2 > SIMD_EPS * SIMD_EPS) {
            tn = vdiv_bt(tn, sqrtf(l2));
            const float distance2 = -len(pd.wa - pd.wb);
            if (!p.valid || distance2 < p.distance) { p.distance = distance2; p.pa = pd.wa; p.pb = pd.wb; p.normal = tn; p.valid = true; }   // only replace valid penetrations when the result is deeper
        }
    } else if (len2(axis) > 0.f) {
        // EPA reports no penetration and the second GJK (cores, no margins) a positive distance (:894-921)
        const float distance2 = len(pd.wa - pd.wb) - margin;
        if (!p.valid || distance2 < p.distance) {
            p.distance = distance2; p.pa = pd.wa; p.pb = pd.wb;
            p.pa -= axis * margin_a; p.pb += axis * margin_b;
            p.normal = normalized(axis);
            p.valid = true;
        }
    }
    return true;
}

// One (hitbox, triangle) pair.  bc / R: the hitbox child's world transform; core: btBoxShape's implicit dimensions; margin_a: its
// collision margin; the triangle has margin 0 (btConcaveShape.cpp:21) and sits in a body at the origin.  `breaking`: the manifold's
// contact breaking threshold.  true: `out` is the point btManifoldResult::addContactPoint receives.
// (shape B in general: up to three vertices t0..t2 in the frame of a body at origin_b, collision margin margin_b -- the mesh triangle
// is (origin 0, margin 0), the ball's btSphereShape is the single point (0,0,0) with the radius as its margin, btSphereShape.h:41-52)
RLG_HD bool gjk_box_convex(V3 bc, const M3& R, V3 core, float margin_a, V3 origin_b, V3 t0, V3 t1, V3 t2, float margin_b, float breaking, GjkOut& out, bool& deep) {
    deep = false;
    const V3 offset = (bc + origin_b) * 0.5f;                    // positionOffset
    GjkShapes sh;
    sh.R = R; sh.core = core; sh.oa = bc - offset; sh.ob = origin_b - offset;   // local origins
    sh.t0 = t0; sh.t1 = t1; sh.t2 = t2;
    const V3 oa = sh.oa, ob = sh.ob;
    const float margin = margin_a + margin_b;
    float max_d2 = margin_a + margin_b + breaking; max_d2 *= max_d2;    // btConvexConvexAlgorithm.cpp:313-317
    V3 axis = v3(0, 1, 0);
    GjkSimplex s; s.n = 0; s.codes = 0u; s.needs_update = true; s.valid = false; s.degenerate = false;
    s.w0 = s.w1 = s.w2 = s.w3 = v3(0, 0, 0);
    s.last_w = v3(1e18f, 1e18f, 1e18f); s.cp1 = s.cp2 = s.cv = v3(0, 0, 0);
    s.bc0 = s.bc1 = s.bc2 = s.bc3 = 0.f; s.used = 0u;
    float sq_dist = 1e18f;
    int degenerate = 0; bool check_simplex = false;
    int iter = 0;
    for (;;) {
        V3 dir_a = tmul(R, -axis);                                  // (-axis) * basisA
        V3 dir_b = axis;                                            // axis * identity
        float d0 = dot(dir_b, sh.t0), d1 = dot(dir_b, sh.t1), d2 = dot(dir_b, sh.t2);
        int mi = d0 < d1 ? (d1 < d2 ? 2 : 1) : (d0 < d2 ? 2 : 0);  // btVector3::maxAxis
        const uint32_t code = (dir_a.x >= 0.f ? 1u : 0u) | (dir_a.y >= 0.f ? 2u : 0u) | (dir_a.z >= 0.f ? 4u : 0u) | ((uint32_t)mi << 3);
        V3 pw = sh.point_a(code);                                   // (R * localGetSupportVertexWithoutMargin) + origin
        V3 qw = sh.point_b(code);
        V3 w = pw - qw;
        float delta = dot(axis, w);
        if (delta > 0.f && delta * delta > sq_dist * max_d2) { degenerate = 10; check_simplex = true; break; }
        if (gjk_in_simplex(s, w)) { degenerate = 1; check_simplex = true; break; }
        float f0 = sq_dist - delta, f1 = sq_dist * GJK_REL_ERROR2;
        if (f0 <= f1) { degenerate = f0 <= 0.f ? 2 : 11; check_simplex = true; break; }
        s.last_w = w; s.needs_update = true;
        gjk_append(s, w, code);
#ifdef RLG_GJK_STATS
        RLG_GJK_STATS(4 + s.n, 0);
#endif
        if (!gjk_update(s, sh)) { degenerate = 3; check_simplex = true; break; }
        V3 nv = s.cv;
        if (len2(nv) < GJK_REL_ERROR2) { axis = nv; degenerate = 6; check_simplex = true; break; }
        float prev = sq_dist;
        sq_dist = len2(nv);
        if (prev - sq_dist <= SIMD_EPS * prev) { check_simplex = true; degenerate = 12; break; }
        axis = nv;
        if (iter++ > 1000) break;
        if (s.n == 4) { degenerate = 13; break; }
    }
#ifdef RLG_GJK_STATS
    RLG_GJK_STATS(2, iter);   // runs, and their iteration counts (sum / max / histogram)
#endif
    bool valid = false; float distance = 0.f; V3 normal = v3(0, 0, 0), pa = v3(0, 0, 0), pb = v3(0, 0, 0);
    if (check_simplex) {
        gjk_update(s, sh);
        pa = s.cp1; pb = s.cp2;
        normal = axis;
        float l2 = len2(axis);
        if (l2 < 0.0001f) degenerate = 5;
        if (l2 > SIMD_EPS * SIMD_EPS) {
            float rlen = 1.f / sqrtf(l2);
            normal *= rlen;
            float sd = sqrtf(sq_dist);
            pa -= axis * (margin_a / sd);
            pb += axis * (margin_b / sd);
            distance = (1.f / rlen) - margin;
            valid = true;
        }
    }
    // btGjkPairDetector.cpp:847-927: penetration, or a degenerate ending with the cores closer than 0.01: the reference asks its penetration
    // depth solver (a second GJK + EPA on the margin-inflated shapes, arena_epa.h) and keeps whichever answer is deeper
    if (!valid || (degenerate && (distance + margin) < 0.01f)) {
#ifdef RLG_EXPERIMENT_NO_EPA   // what-if build only: round 2's behaviour (the caller's minimum-translation stand-in), to price the solver's presence
        deep = true; if (!valid) return false;
#else
        GjkPen pen; pen.valid = valid; pen.distance = distance; pen.pa = pa; pen.pb = pb; pen.normal = normal;
        if (!gjk_penetration(R, oa, core, margin_a, ob, t0, t1, t2, margin_b, pen)) { deep = true; return false; }   // (no full-size arena: the caller's minimum-translation answer stands in, counted)
        valid = pen.valid; distance = pen.distance; pa = pen.pa; pb = pen.pb; normal = pen.normal;
        if (!valid) return false;
#endif
    }
    if (!(distance < 0.f || distance * distance < max_d2)) return false;
    {   // m_fixContactNormalDirection (:929-948): the normal must point from the triangle's box centre towards the hitbox's
        V3 e = abs_rows_dot(R, v3(core.x + margin_a, core.y + margin_a, core.z + margin_a));
        V3 amin = oa - e, amax = oa + e;
        V3 pos_a = (amax + amin) * 0.5f;
        V3 bmin = vmin(vmin(sh.t0 + ob, sh.t1 + ob), sh.t2 + ob) - v3(margin_b, margin_b, margin_b), bmax = vmax(vmax(sh.t0 + ob, sh.t1 + ob), sh.t2 + ob) + v3(margin_b, margin_b, margin_b);
        V3 pos_b = (bmin + bmax) * 0.5f;
        if (dot(pos_a - pos_b, normal) < 0.f) normal *= -1.f;
    }
    out.n = normal; out.pb = pb + offset; out.dist = distance;
    return true;
}

// the two out-of-line instances (the general form is inlined into each: as a call its eighteen by-value floats would travel through the stack)
RLG_HD_T5 bool gjk_box_triangle(V3 bc, const M3& R, V3 core, float margin_a, const MeshTri& t, float breaking, GjkOut& out, bool& deep) {
    return gjk_box_convex(bc, R, core, margin_a, v3(0, 0, 0), v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z), 0.f, breaking, out, deep);
}
RLG_HD_T7 bool gjk_box_sphere(V3 bc, const M3& R, V3 core, float margin_a, V3 centre, float radius, float breaking, GjkOut& out, bool& deep) {
    return gjk_box_convex(bc, R, core, margin_a, centre, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), radius, breaking, out, deep);
}

}  // namespace rlg
