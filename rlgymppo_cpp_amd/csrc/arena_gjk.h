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
#include "arena_epa.h"

#ifndef RLG_GJK_TRIANGLE_FN
#define RLG_GJK_TRIANGLE_FN RLG_HD   /* out of line (one copy for the 3-vertex case and the tetrahedron's faces) measured slower: 26 K vs 21.5 K cycles per run in isolation */
#endif
#ifndef RLG_GJK_FACE_LOOP
#define RLG_GJK_FACE_LOOP RLG_UNROLL
#endif

// Where the penetration-depth solver keeps its state (arena_epa.h).  Host default: a full-size arena on the stack.  The device kernels
// define these before including this header (rlgpu_env.hip): a small arena in LDS shared by the wavefront's lanes one at a time, and a
// full-size one in global memory for the queries that do not fit.
#ifndef RLG_EPA_ARENA_DECL
#define RLG_EPA_ARENA_DECL alignas(16) unsigned char epa_mem_[epa_arena_bytes(EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES)]; \
    EpaArena epa_small_ = epa_arena_at(epa_mem_, EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); EpaArena* epa_big_ = nullptr;
#define RLG_EPA_SERIALIZE_BEGIN
#define RLG_EPA_SERIALIZE_END
#define RLG_EPA_COUNT_BIG() ((void)0)
#endif

namespace rlg {

// The two shapes of one query.  A support point is a box corner (3 sign bits) or a triangle vertex (2 bits): the simplex remembers
// that code per vertex instead of the two support points (24 floats), and rebuilds them -- with the expressions that produced them, so
// bit for bit -- where btVoronoiSimplexSolver reads its m_simplexPointsP / Q arrays.
// Selections between vectors go through scalar prvalues: `c ? a : b` on two V3 lvalues is itself an lvalue -- the compiler selects the
// ADDRESS and copies from it, and an object read through a computed address cannot be kept in registers.
RLG_HD float gjk_fsel(bool c, float a, float b) { return c ? a : b; }
RLG_HD V3 gjk_sel(bool c, V3 a, V3 b) { return v3(gjk_fsel(c, a.x, b.x), gjk_fsel(c, a.y, b.y), gjk_fsel(c, a.z, b.z)); }

struct GjkShapes {
    M3 R; V3 core; V3 oa, ob; V3 t0, t1, t2;
    RLG_HD V3 point_a(uint32_t code) const { return (R * v3((code & 1u) ? core.x : -core.x, (code & 2u) ? core.y : -core.y, (code & 4u) ? core.z : -core.z)) + oa; }
    RLG_HD V3 point_b(uint32_t code) const { const uint32_t k = (code >> 3) & 3u; return gjk_sel(k == 0, t0, gjk_sel(k == 1, t1, t2)) + ob; }
};

struct GjkSimplex {
    V3 w0, w1, w2, w3;      // Minkowski points
    uint32_t codes;         // 5 bits per vertex: which corner of A, which vertex of B
    int n;
    V3 last_w;
    V3 cp1, cp2, cv;        // cached closest points on A / B and their difference
    float bc0, bc1, bc2, bc3;   // barycentric coordinates of the closest point
    uint32_t used;          // bit k: vertex k supports the closest point
    bool degenerate, needs_update, valid;
};

// (No slot of the simplex is ever addressed with a run-time index -- appends, the "move the last vertex into the hole" of
// removeVertex and the face loop of the tetrahedron case go through compare chains -- so that on the device all of it stays in
// registers: indexed by s.n it sat in scratch memory, and a hitbox-triangle item was ~5x the cycles of the SAT routine it replaced.)
RLG_HD V3 gjk_w(const GjkSimplex& s, int k) { return gjk_sel(k == 0, s.w0, gjk_sel(k == 1, s.w1, gjk_sel(k == 2, s.w2, s.w3))); }
RLG_HD uint32_t gjk_code(const GjkSimplex& s, int k) { return (s.codes >> (5 * k)) & 31u; }
RLG_HD void gjk_set_slot(GjkSimplex& s, int k, V3 w, uint32_t code) {
    // every slot is assigned, by value: conditional stores would be merged into one store through a selected POINTER, and an object
    // addressed that way stays in (scratch) memory (gjk_sel: likewise for reads)
    s.w0 = gjk_sel(k == 0, w, s.w0); s.w1 = gjk_sel(k == 1, w, s.w1); s.w2 = gjk_sel(k == 2, w, s.w2); s.w3 = gjk_sel(k == 3, w, s.w3);
    s.codes = (s.codes & ~(31u << (5 * k))) | (code << (5 * k));
}
RLG_HD void gjk_remove_vertex(GjkSimplex& s, int k) { s.n--; gjk_set_slot(s, k, gjk_w(s, s.n), gjk_code(s, s.n)); }   // removeVertex: the last one fills the hole
RLG_HD void gjk_append(GjkSimplex& s, V3 w, uint32_t code) { gjk_set_slot(s, s.n, w, code); s.n++; }
RLG_HD void gjk_reduce(GjkSimplex& s) {   // btVoronoiSimplexSolver::reduceVertices
    if (s.n >= 4 && !(s.used & 8u)) gjk_remove_vertex(s, 3);
    if (s.n >= 3 && !(s.used & 4u)) gjk_remove_vertex(s, 2);
    if (s.n >= 2 && !(s.used & 2u)) gjk_remove_vertex(s, 1);
    if (s.n >= 1 && !(s.used & 1u)) gjk_remove_vertex(s, 0);
}
struct GjkSub { V3 closest; float b0, b1, b2; uint32_t used; };   // closest point of one triangle: barycentrics and support bits of its 3 vertices

// btVoronoiSimplexSolver::closestPtPointTriangle with p = origin (btVoronoiSimplexSolver.cpp:313-405)
RLG_GJK_TRIANGLE_FN GjkSub gjk_origin_triangle(V3 a, V3 b, V3 c) {
    GjkSub r;
    const V3 p = v3(0, 0, 0);
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { r.closest = a; r.used = 1u; r.b0 = 1; r.b1 = 0; r.b2 = 0; return r; }
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { r.closest = b; r.used = 2u; r.b0 = 0; r.b1 = 1; r.b2 = 0; return r; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        float v = d1 / (d1 - d3);
        r.closest = a + v * ab; r.used = 3u; r.b0 = 1 - v; r.b1 = v; r.b2 = 0; return r;
    }
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { r.closest = c; r.used = 4u; r.b0 = 0; r.b1 = 0; r.b2 = 1; return r; }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        float w = d2 / (d2 - d6);
        r.closest = a + w * ac; r.used = 5u; r.b0 = 1 - w; r.b1 = 0; r.b2 = w; return r;
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        r.closest = b + w * (c - b); r.used = 6u; r.b0 = 0; r.b1 = 1 - w; r.b2 = w; return r;
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    r.closest = a + ab * v + ac * w;
    r.used = 7u; r.b0 = 1 - v - w; r.b1 = v; r.b2 = w;
    return r;
}
// pointOutsideOfPlane with p = origin: 1 outside, 0 inside, -1 degenerate tetrahedron (:408-434)
RLG_HD int gjk_origin_outside(V3 a, V3 b, V3 c, V3 d) {
    V3 normal = cross(b - a, c - a);
    float signp = dot(v3(0, 0, 0) - a, normal), signd = dot(d - a, normal);
    if (signd * signd < (1e-4f * 1e-4f)) return -1;
    return (signp * signd < 0.f) ? 1 : 0;
}
// closestPtPointTetrahedron with p = origin (:436-577) on the simplex' four points; writes the simplex' barycentrics and support bits.
// false: the origin is inside (or the tetrahedron is degenerate).  The four faces in the reference's order -- abc, acd, adb, bdc, each
// tested against the remaining vertex d, b, c, a -- are walked by ONE copy of the face code: 2-bit vertex numbers per face.
RLG_HD bool gjk_origin_tetrahedron(GjkSimplex& s, V3& closest, bool& degenerate) {
    constexpr uint32_t FACE_I = 0u | (0u << 2) | (0u << 4) | (1u << 6), FACE_J = 1u | (2u << 2) | (3u << 4) | (3u << 6),
                       FACE_K = 2u | (3u << 2) | (1u << 4) | (2u << 6), FACE_OPP = 3u | (1u << 2) | (2u << 4) | (0u << 6);
    closest = v3(0, 0, 0);
    s.used = 15u;
    uint32_t outside = 0; bool bad = false;
    RLG_GJK_FACE_LOOP
    for (int f = 0; f < 4; f++) {
        const int i = (FACE_I >> (2 * f)) & 3, j = (FACE_J >> (2 * f)) & 3, k = (FACE_K >> (2 * f)) & 3, o = (FACE_OPP >> (2 * f)) & 3;
        const int side = gjk_origin_outside(gjk_w(s, i), gjk_w(s, j), gjk_w(s, k), gjk_w(s, o));
        bad = bad || side < 0;
        if (side > 0) outside |= 1u << f;
    }
    if (bad) { degenerate = true; return false; }
    if (!outside) return false;
    float best = 3.402823466e+38f;
    RLG_GJK_FACE_LOOP
    for (int f = 0; f < 4; f++) {
        if (!((outside >> f) & 1u)) continue;
        const int i = (FACE_I >> (2 * f)) & 3, j = (FACE_J >> (2 * f)) & 3, k = (FACE_K >> (2 * f)) & 3;
        const GjkSub t = gjk_origin_triangle(gjk_w(s, i), gjk_w(s, j), gjk_w(s, k));
        const float sq = dot(t.closest, t.closest);
        if (sq < best) {
            best = sq; closest = t.closest;
            s.used = ((t.used & 1u) ? (1u << i) : 0u) | ((t.used & 2u) ? (1u << j) : 0u) | ((t.used & 4u) ? (1u << k) : 0u);
            s.bc0 = i == 0 ? t.b0 : (j == 0 ? t.b1 : (k == 0 ? t.b2 : 0.f));
            s.bc1 = i == 1 ? t.b0 : (j == 1 ? t.b1 : (k == 1 ? t.b2 : 0.f));
            s.bc2 = i == 2 ? t.b0 : (j == 2 ? t.b1 : (k == 2 ? t.b2 : 0.f));
            s.bc3 = i == 3 ? t.b0 : (j == 3 ? t.b1 : (k == 3 ? t.b2 : 0.f));
        }
    }
    return true;
}
RLG_HD bool gjk_bc_valid(const GjkSimplex& s) { return s.bc0 >= 0.f && s.bc1 >= 0.f && s.bc2 >= 0.f && s.bc3 >= 0.f; }

// btVoronoiSimplexSolver::updateClosestVectorAndPoints (:81-237)
RLG_HD bool gjk_update(GjkSimplex& s, const GjkShapes& sh) {
    if (!s.needs_update) return s.valid;
    s.needs_update = false;
    s.degenerate = false;
    s.bc0 = s.bc1 = s.bc2 = s.bc3 = 0.f;
    s.used = 0u;
    if (s.n == 1) {
        s.cp1 = sh.point_a(gjk_code(s, 0)); s.cp2 = sh.point_b(gjk_code(s, 0)); s.cv = s.cp1 - s.cp2;
        s.bc0 = 1.f;
        s.valid = gjk_bc_valid(s);
    } else if (s.n == 2) {
        const V3 from = s.w0, to = s.w1;
        V3 diff = v3(0, 0, 0) - from, v = to - from;
        float t = dot(v, diff);
        if (t > 0.f) {
            float dvv = dot(v, v);
            if (t < dvv) { t /= dvv; diff -= t * v; s.used = 3u; }
            else { t = 1.f; diff -= v; s.used = 2u; }
        } else { t = 0.f; s.used = 1u; }
        s.bc0 = 1 - t; s.bc1 = t;
        const V3 p0 = sh.point_a(gjk_code(s, 0)), p1 = sh.point_a(gjk_code(s, 1)), q0 = sh.point_b(gjk_code(s, 0)), q1 = sh.point_b(gjk_code(s, 1));
        s.cp1 = p0 + t * (p1 - p0);
        s.cp2 = q0 + t * (q1 - q0);
        s.cv = s.cp1 - s.cp2;
        gjk_reduce(s);
        s.valid = gjk_bc_valid(s);
    } else if (s.n == 3) {
        const GjkSub r = gjk_origin_triangle(s.w0, s.w1, s.w2);
        s.bc0 = r.b0; s.bc1 = r.b1; s.bc2 = r.b2; s.bc3 = 0.f;
        s.used = r.used;
        s.cp1 = sh.point_a(gjk_code(s, 0)) * s.bc0 + sh.point_a(gjk_code(s, 1)) * s.bc1 + sh.point_a(gjk_code(s, 2)) * s.bc2;
        s.cp2 = sh.point_b(gjk_code(s, 0)) * s.bc0 + sh.point_b(gjk_code(s, 1)) * s.bc1 + sh.point_b(gjk_code(s, 2)) * s.bc2;
        s.cv = s.cp1 - s.cp2;
        gjk_reduce(s);
        s.valid = gjk_bc_valid(s);
    } else if (s.n == 4) {
        bool deg = false; V3 closest;
        const bool sep = gjk_origin_tetrahedron(s, closest, deg);
        s.degenerate = deg;
        if (sep) {
            s.cp1 = sh.point_a(gjk_code(s, 0)) * s.bc0 + sh.point_a(gjk_code(s, 1)) * s.bc1 + sh.point_a(gjk_code(s, 2)) * s.bc2 + sh.point_a(gjk_code(s, 3)) * s.bc3;
            s.cp2 = sh.point_b(gjk_code(s, 0)) * s.bc0 + sh.point_b(gjk_code(s, 1)) * s.bc1 + sh.point_b(gjk_code(s, 2)) * s.bc2 + sh.point_b(gjk_code(s, 3)) * s.bc3;
            s.cv = s.cp1 - s.cp2;
            gjk_reduce(s);
            s.valid = gjk_bc_valid(s);
        } else if (deg) s.valid = false;
        else { s.valid = true; s.cv = v3(0, 0, 0); }
    } else s.valid = false;
    return s.valid;
}
RLG_HD bool v3_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

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
        {   // inSimplex
            bool found = false;
            if (s.n > 0 && v3_eq(s.w0, w)) found = true;
            if (s.n > 1 && v3_eq(s.w1, w)) found = true;
            if (s.n > 2 && v3_eq(s.w2, w)) found = true;
            if (s.n > 3 && v3_eq(s.w3, w)) found = true;
            if (v3_eq(w, s.last_w)) found = true;
            if (found) { degenerate = 1; check_simplex = true; break; }
        }
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
        GjkPen pen; pen.valid = valid; pen.distance = distance; pen.pa = pa; pen.pb = pb; pen.normal = normal;
        if (!gjk_penetration(R, oa, core, margin_a, ob, t0, t1, t2, margin_b, pen)) { deep = true; return false; }   // (no full-size arena: the caller's minimum-translation answer stands in, counted)
        valid = pen.valid; distance = pen.distance; pa = pen.pa; pb = pen.pb; normal = pen.normal;
        if (!valid) return false;
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
RLG_HD_NOINLINE bool gjk_box_triangle(V3 bc, const M3& R, V3 core, float margin_a, const MeshTri& t, float breaking, GjkOut& out, bool& deep) {
    return gjk_box_convex(bc, R, core, margin_a, v3(0, 0, 0), v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z), 0.f, breaking, out, deep);
}
RLG_HD_NOINLINE bool gjk_box_sphere(V3 bc, const M3& R, V3 core, float margin_a, V3 centre, float radius, float breaking, GjkOut& out, bool& deep) {
    return gjk_box_convex(bc, R, core, margin_a, centre, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), radius, breaking, out, deep);
}

}  // namespace rlg
