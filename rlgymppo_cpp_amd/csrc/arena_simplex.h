// arena_simplex.h — btVoronoiSimplexSolver restated (NarrowPhaseCollision/btVoronoiSimplexSolver.cpp): the simplex bookkeeping shared by the
// hitbox-triangle GJK (arena_gjk.h) and the wheel rays' convex cast against other cars and the ball (arena_world.h:ray_convex_cast).
// Split out of arena_gjk.h; see there for why the simplex is laid out the way it is (no run-time indexed slots: registers on the device).
//
// Bullet (zlib licence): Bullet Continuous Collision Detection and Physics Library, Copyright (c) 2003-2006 Erwin Coumans.  The closest-point
// routines follow Christer Ericson's Real-Time Collision Detection as Bullet does.  This file is an altered restatement, not the original.
#pragma once
#include "rl_math.h"

#ifndef RLG_GJK_TRIANGLE_FN
#define RLG_GJK_TRIANGLE_FN RLG_HD   /* out of line (one copy for the 3-vertex case and the tetrahedron's faces) measured slower: 26 K vs 21.5 K cycles per run in isolation */
#endif
#ifndef RLG_GJK_FACE_LOOP
#define RLG_GJK_FACE_LOOP RLG_UNROLL
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
template <class SH>
RLG_HD bool gjk_update(GjkSimplex& s, const SH& sh) {
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
// btVoronoiSimplexSolver::inSimplex (btVoronoiSimplexSolver.cpp:267-292; BT_USE_EQUAL_VERTEX_THRESHOLD is defined, btVoronoiSimplexSolver.h:24):
// w is within sqrt(1e-4) of a vertex of the current simplex, or equals the vertex added last
constexpr float GJK_EQUAL_VERTEX_THRESHOLD = 0.0001f;
RLG_HD bool gjk_in_simplex(const GjkSimplex& s, V3 w) {
    bool found = false;
    if (s.n > 0 && len2(w - s.w0) <= GJK_EQUAL_VERTEX_THRESHOLD) found = true;
    if (s.n > 1 && len2(w - s.w1) <= GJK_EQUAL_VERTEX_THRESHOLD) found = true;
    if (s.n > 2 && len2(w - s.w2) <= GJK_EQUAL_VERTEX_THRESHOLD) found = true;
    if (s.n > 3 && len2(w - s.w3) <= GJK_EQUAL_VERTEX_THRESHOLD) found = true;
    if (v3_eq(w, s.last_w)) found = true;
    return found;
}
RLG_HD void gjk_reset(GjkSimplex& s) {   // btVoronoiSimplexSolver::reset
    s.n = 0; s.codes = 0u; s.needs_update = true; s.valid = false; s.degenerate = false;
    s.w0 = s.w1 = s.w2 = s.w3 = v3(0, 0, 0);
    s.last_w = v3(1e18f, 1e18f, 1e18f); s.cp1 = s.cp2 = s.cv = v3(0, 0, 0);
    s.bc0 = s.bc1 = s.bc2 = s.bc3 = 0.f; s.used = 0u;
}

// ---- btSubsimplexConvexCast::calcTimeOfImpact (NarrowPhaseCollision/btSubSimplexConvexCast.cpp:31-142) as btCollisionWorld::
// rayTestSingleInternal sets it up for a ray against a convex shape (CollisionDispatch/btCollisionWorld.cpp:267-310): shape A is a point
// (btSphereShape of radius 0 and margin 0) travelling from `from` to `to`, shape B stands still at (Rb, ob) and answers with
// localGetSupportingVertex -- the sharp box (btBoxShape.h: half extents incl. margin) or a sphere (btSphereShape.cpp: radius * normalised
// direction).  The simplex keeps its support points by value (five ids: the solver's arrays hold VORONOI_SIMPLEX_MAX_VERTS = 5).
struct CastPoints {
    V3 p0, p1, p2, p3, p4, q0, q1, q2, q3, q4;
    RLG_HD V3 point_a(uint32_t id) const { return gjk_sel(id == 0, p0, gjk_sel(id == 1, p1, gjk_sel(id == 2, p2, gjk_sel(id == 3, p3, p4)))); }
    RLG_HD V3 point_b(uint32_t id) const { return gjk_sel(id == 0, q0, gjk_sel(id == 1, q1, gjk_sel(id == 2, q2, gjk_sel(id == 3, q3, q4)))); }
    RLG_HD void set(uint32_t id, V3 p, V3 q) {
        p0 = gjk_sel(id == 0, p, p0); p1 = gjk_sel(id == 1, p, p1); p2 = gjk_sel(id == 2, p, p2); p3 = gjk_sel(id == 3, p, p3); p4 = gjk_sel(id == 4, p, p4);
        q0 = gjk_sel(id == 0, q, q0); q1 = gjk_sel(id == 1, q, q1); q2 = gjk_sel(id == 2, q, q2); q3 = gjk_sel(id == 3, q, q3); q4 = gjk_sel(id == 4, q, q4);
    }
};
RLG_HD V3 cast_support_b(const M3& Rb, V3 origin, V3 half, float radius, V3 dir_world) {
    const V3 dl = tmul(Rb, dir_world);                                  // v * basis
    V3 sup;
    if (radius > 0.f) {
        V3 n = dl;
        if (len2(n) < SIMD_EPS * SIMD_EPS) n = v3(-1.f, -1.f, -1.f);
        n = normalized(n);
        sup = v3(0.f, 0.f, 0.f) + n * radius;                            // btSphereShape::localGetSupportingVertex
    } else sup = v3(dl.x >= 0.f ? half.x : -half.x, dl.y >= 0.f ? half.y : -half.y, dl.z >= 0.f ? half.z : -half.z);   // btBoxShape::localGetSupportingVertex
    return (Rb * sup) + origin;
}
RLG_HD V3 interp3(V3 v0, V3 v1, float rt) { const float s = 1.0f - rt; return v3(s * v0.x + rt * v1.x, s * v0.y + rt * v1.y, s * v0.z + rt * v1.z); }   // btVector3::setInterpolate3
RLG_HD_NOINLINE bool ray_convex_cast(V3 from, V3 to, const M3& Rb, V3 ob, V3 half, float radius, float& fraction, V3& normal) {
    GjkSimplex s; gjk_reset(s);
    CastPoints cp; cp.p0 = cp.p1 = cp.p2 = cp.p3 = cp.p4 = cp.q0 = cp.q1 = cp.q2 = cp.q3 = cp.q4 = v3(0, 0, 0);
    const V3 r = (to - from) - (ob - ob);                                // linVelA - linVelB
    float lambda = 0.f;
    V3 ia = from, ib = ob;                                               // the interpolated transforms' origins
    V3 sup_a = from;                                                     // fromA(point support) = the origin: the point shape's support is 0 + 0 * n
    V3 sup_b = cast_support_b(Rb, ob, half, radius, r);
    V3 v = sup_a - sup_b;
    int max_iter = 32;                                                   // m_subSimplexCastMaxIterations
    V3 n = v3(0, 0, 0);
    float dist2 = len2(v);
    while ((dist2 > 0.0001f) && max_iter--) {                            // m_subSimplexCastEpsilon
        sup_a = ia;
        sup_b = cast_support_b(Rb, ib, half, radius, v);
        V3 w = sup_a - sup_b;
        const float VdotW = dot(v, w);
        if (lambda > 1.0f) return false;
        if (VdotW > 0.f) {
            const float VdotR = dot(v, r);
            if (VdotR >= -(SIMD_EPS * SIMD_EPS)) return false;
            lambda = lambda - VdotW / VdotR;
            ia = interp3(from, to, lambda);
            ib = interp3(ob, ob, lambda);
            w = sup_a - sup_b;
            n = v;
        }
        if (!gjk_in_simplex(s, w)) {
            // addVertex: an id none of the live vertices uses
            uint32_t usedm = 0u;
            for (int k = 0; k < 4; k++) if (k < s.n) usedm |= 1u << gjk_code(s, k);
            uint32_t id = 0; while (id < 4u && ((usedm >> id) & 1u)) id++;
            cp.set(id, sup_a, sup_b);
            s.last_w = w; s.needs_update = true;
            gjk_append(s, w, id);
        }
        const bool ok = gjk_update(s, cp);
        v = s.cv;
        if (ok) dist2 = len2(v); else dist2 = 0.f;
    }
    fraction = lambda;
    if (len2(n) >= SIMD_EPS * SIMD_EPS) normal = normalized(n); else normal = v3(0, 0, 0);
    if (dot(normal, r) >= -0.f) return false;                            // m_allowedPenetration = 0
    return true;
}


}  // namespace rlg
