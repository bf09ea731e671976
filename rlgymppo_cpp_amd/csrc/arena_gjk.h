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
// Penetration deeper than the margin (cores overlap; the reference then runs EPA on the margin-inflated shapes,
// btGjkEpaPenetrationDepthSolver.cpp) is answered by the minimum-translation axis of the two core polytopes plus the margin:
// same depth and normal as a converged EPA, a witness point that may differ where the deepest feature is not a single point.
#pragma once
#include "arena_world.h"

namespace rlg {

struct GjkSimplex {
    V3 w[4], p[4], q[4];    // Minkowski point, support point on A, on B
    int n;
    V3 last_w;
    V3 cp1, cp2, cv;        // cached closest points on A / B and their difference
    float bc[4];            // barycentric coordinates of the closest point
    bool used[4];
    bool degenerate, needs_update, valid;
};

// (Every slot of the simplex is addressed with compile-time indices -- appends and the "move the last vertex into the hole" of
// removeVertex go through compare chains on the count -- so that on the device the 36 floats stay in registers: indexed by s.n they
// sat in scratch memory, and a hitbox-triangle item was ~5x the cycles of the SAT routine it replaced.)
RLG_HD V3 gjk_pick(const V3 (&a)[4], int k) { return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3])); }
template <int I>
RLG_HD void gjk_remove_vertex(GjkSimplex& s) { s.n--; const int k = s.n; s.w[I] = gjk_pick(s.w, k); s.p[I] = gjk_pick(s.p, k); s.q[I] = gjk_pick(s.q, k); }
RLG_HD void gjk_append(GjkSimplex& s, V3 w, V3 p, V3 q) {
    if (s.n == 0) { s.w[0] = w; s.p[0] = p; s.q[0] = q; }
    else if (s.n == 1) { s.w[1] = w; s.p[1] = p; s.q[1] = q; }
    else if (s.n == 2) { s.w[2] = w; s.p[2] = p; s.q[2] = q; }
    else { s.w[3] = w; s.p[3] = p; s.q[3] = q; }
    s.n++;
}
RLG_HD void gjk_reduce(GjkSimplex& s) {   // btVoronoiSimplexSolver::reduceVertices
    if (s.n >= 4 && !s.used[3]) gjk_remove_vertex<3>(s);
    if (s.n >= 3 && !s.used[2]) gjk_remove_vertex<2>(s);
    if (s.n >= 2 && !s.used[1]) gjk_remove_vertex<1>(s);
    if (s.n >= 1 && !s.used[0]) gjk_remove_vertex<0>(s);
}
struct GjkSub { V3 closest; float bc[4]; bool used[4]; };
RLG_HD void gjk_sub_set(GjkSub& r, float a, float b, float c, float d) { r.bc[0] = a; r.bc[1] = b; r.bc[2] = c; r.bc[3] = d; }
RLG_HD void gjk_sub_used(GjkSub& r, bool a, bool b, bool c, bool d) { r.used[0] = a; r.used[1] = b; r.used[2] = c; r.used[3] = d; }

// btVoronoiSimplexSolver::closestPtPointTriangle with p = origin (btVoronoiSimplexSolver.cpp:313-405)
RLG_HD void gjk_origin_triangle(V3 a, V3 b, V3 c, GjkSub& r) {
    gjk_sub_used(r, false, false, false, false);
    const V3 p = v3(0, 0, 0);
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { r.closest = a; r.used[0] = true; gjk_sub_set(r, 1, 0, 0, 0); return; }
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { r.closest = b; r.used[1] = true; gjk_sub_set(r, 0, 1, 0, 0); return; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        float v = d1 / (d1 - d3);
        r.closest = a + v * ab; r.used[0] = true; r.used[1] = true; gjk_sub_set(r, 1 - v, v, 0, 0); return;
    }
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { r.closest = c; r.used[2] = true; gjk_sub_set(r, 0, 0, 1, 0); return; }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        float w = d2 / (d2 - d6);
        r.closest = a + w * ac; r.used[0] = true; r.used[2] = true; gjk_sub_set(r, 1 - w, 0, w, 0); return;
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        r.closest = b + w * (c - b); r.used[1] = true; r.used[2] = true; gjk_sub_set(r, 0, 1 - w, w, 0); return;
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    r.closest = a + ab * v + ac * w;
    gjk_sub_used(r, true, true, true, false);
    gjk_sub_set(r, 1 - v - w, v, w, 0);
}
// pointOutsideOfPlane with p = origin: 1 outside, 0 inside, -1 degenerate tetrahedron (:408-434)
RLG_HD int gjk_origin_outside(V3 a, V3 b, V3 c, V3 d) {
    V3 normal = cross(b - a, c - a);
    float signp = dot(v3(0, 0, 0) - a, normal), signd = dot(d - a, normal);
    if (signd * signd < (1e-4f * 1e-4f)) return -1;
    return (signp * signd < 0.f) ? 1 : 0;
}
// closestPtPointTetrahedron with p = origin (:436-577).  false: the origin is inside (or the tetrahedron is degenerate)
RLG_HD bool gjk_origin_tetrahedron(V3 a, V3 b, V3 c, V3 d, GjkSub& fin, bool& degenerate) {
    fin.closest = v3(0, 0, 0);
    gjk_sub_used(fin, true, true, true, true);
    int oabc = gjk_origin_outside(a, b, c, d), oacd = gjk_origin_outside(a, c, d, b), oadb = gjk_origin_outside(a, d, b, c), obdc = gjk_origin_outside(b, d, c, a);
    if (oabc < 0 || oacd < 0 || oadb < 0 || obdc < 0) { degenerate = true; return false; }
    if (!oabc && !oacd && !oadb && !obdc) return false;
    float best = 3.402823466e+38f;
    GjkSub t;
    if (oabc) {
        gjk_origin_triangle(a, b, c, t);
        float sq = dot(t.closest, t.closest);
        if (sq < best) { best = sq; fin.closest = t.closest; gjk_sub_used(fin, t.used[0], t.used[1], t.used[2], false); gjk_sub_set(fin, t.bc[0], t.bc[1], t.bc[2], 0); }
    }
    if (oacd) {
        gjk_origin_triangle(a, c, d, t);
        float sq = dot(t.closest, t.closest);
        if (sq < best) { best = sq; fin.closest = t.closest; gjk_sub_used(fin, t.used[0], false, t.used[1], t.used[2]); gjk_sub_set(fin, t.bc[0], 0, t.bc[1], t.bc[2]); }
    }
    if (oadb) {
        gjk_origin_triangle(a, d, b, t);
        float sq = dot(t.closest, t.closest);
        if (sq < best) { best = sq; fin.closest = t.closest; gjk_sub_used(fin, t.used[0], t.used[2], false, t.used[1]); gjk_sub_set(fin, t.bc[0], t.bc[2], 0, t.bc[1]); }
    }
    if (obdc) {
        gjk_origin_triangle(b, d, c, t);
        float sq = dot(t.closest, t.closest);
        if (sq < best) { best = sq; fin.closest = t.closest; gjk_sub_used(fin, false, t.used[0], t.used[2], t.used[1]); gjk_sub_set(fin, 0, t.bc[0], t.bc[2], t.bc[1]); }
    }
    return true;
}
RLG_HD bool gjk_bc_valid(const float* bc) { return bc[0] >= 0.f && bc[1] >= 0.f && bc[2] >= 0.f && bc[3] >= 0.f; }

// btVoronoiSimplexSolver::updateClosestVectorAndPoints (:81-237)
RLG_HD bool gjk_update(GjkSimplex& s) {
    if (!s.needs_update) return s.valid;
    s.needs_update = false;
    s.degenerate = false;
    s.bc[0] = s.bc[1] = s.bc[2] = s.bc[3] = 0.f;
    s.used[0] = s.used[1] = s.used[2] = s.used[3] = false;
    if (s.n == 1) {
        s.cp1 = s.p[0]; s.cp2 = s.q[0]; s.cv = s.cp1 - s.cp2;
        s.bc[0] = 1.f;
        s.valid = gjk_bc_valid(s.bc);
    } else if (s.n == 2) {
        const V3 from = s.w[0], to = s.w[1];
        V3 diff = v3(0, 0, 0) - from, v = to - from;
        float t = dot(v, diff);
        if (t > 0.f) {
            float dvv = dot(v, v);
            if (t < dvv) { t /= dvv; diff -= t * v; s.used[0] = true; s.used[1] = true; }
            else { t = 1.f; diff -= v; s.used[1] = true; }
        } else { t = 0.f; s.used[0] = true; }
        s.bc[0] = 1 - t; s.bc[1] = t;
        s.cp1 = s.p[0] + t * (s.p[1] - s.p[0]);
        s.cp2 = s.q[0] + t * (s.q[1] - s.q[0]);
        s.cv = s.cp1 - s.cp2;
        gjk_reduce(s);
        s.valid = gjk_bc_valid(s.bc);
    } else if (s.n == 3) {
        GjkSub r; gjk_sub_set(r, 0, 0, 0, 0);
        gjk_origin_triangle(s.w[0], s.w[1], s.w[2], r);
        s.bc[0] = r.bc[0]; s.bc[1] = r.bc[1]; s.bc[2] = r.bc[2]; s.bc[3] = r.bc[3];
        s.used[0] = r.used[0]; s.used[1] = r.used[1]; s.used[2] = r.used[2]; s.used[3] = r.used[3];
        s.cp1 = s.p[0] * s.bc[0] + s.p[1] * s.bc[1] + s.p[2] * s.bc[2];
        s.cp2 = s.q[0] * s.bc[0] + s.q[1] * s.bc[1] + s.q[2] * s.bc[2];
        s.cv = s.cp1 - s.cp2;
        gjk_reduce(s);
        s.valid = gjk_bc_valid(s.bc);
    } else if (s.n == 4) {
        GjkSub r; gjk_sub_set(r, 0, 0, 0, 0);
        bool deg = false;
        bool sep = gjk_origin_tetrahedron(s.w[0], s.w[1], s.w[2], s.w[3], r, deg);
        s.bc[0] = r.bc[0]; s.bc[1] = r.bc[1]; s.bc[2] = r.bc[2]; s.bc[3] = r.bc[3];
        s.used[0] = r.used[0]; s.used[1] = r.used[1]; s.used[2] = r.used[2]; s.used[3] = r.used[3];
        s.degenerate = deg;
        if (sep) {
            s.cp1 = s.p[0] * s.bc[0] + s.p[1] * s.bc[1] + s.p[2] * s.bc[2] + s.p[3] * s.bc[3];
            s.cp2 = s.q[0] * s.bc[0] + s.q[1] * s.bc[1] + s.q[2] * s.bc[2] + s.q[3] * s.bc[3];
            s.cv = s.cp1 - s.cp2;
            gjk_reduce(s);
            s.valid = gjk_bc_valid(s.bc);
        } else if (deg) s.valid = false;
        else { s.valid = true; s.cv = v3(0, 0, 0); }
    } else s.valid = false;
    return s.valid;
}
RLG_HD bool v3_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct GjkOut { V3 n, pb; float dist; };
constexpr float GJK_REL_ERROR2 = 1.0e-6f;

// One (hitbox, triangle) pair.  bc / R: the hitbox child's world transform; core: btBoxShape's implicit dimensions; margin_a: its
// collision margin; the triangle has margin 0 (btConcaveShape.cpp:21) and sits in a body at the origin.  `breaking`: the manifold's
// contact breaking threshold.  true: `out` is the point btManifoldResult::addContactPoint receives.
RLG_HD_NOINLINE bool gjk_box_triangle(V3 bc, const M3& R, V3 core, float margin_a, const MeshTri& t, float breaking, GjkOut& out, bool& deep) {
    deep = false;
    const V3 tv[3] = {v3(t.v0x, t.v0y, t.v0z), v3(t.v1x, t.v1y, t.v1z), v3(t.v2x, t.v2y, t.v2z)};
    const V3 offset = (bc + v3(0, 0, 0)) * 0.5f;                 // positionOffset
    const V3 oa = bc - offset, ob = v3(0, 0, 0) - offset;         // local origins
    const float margin = margin_a + 0.f;
    float max_d2 = margin_a + 0.f + breaking; max_d2 *= max_d2;    // btConvexConvexAlgorithm.cpp:313-317
    V3 axis = v3(0, 1, 0);
    GjkSimplex s; s.n = 0; s.needs_update = true; s.valid = false; s.degenerate = false;
    s.last_w = v3(1e18f, 1e18f, 1e18f); s.cp1 = s.cp2 = s.cv = v3(0, 0, 0);
    s.bc[0] = s.bc[1] = s.bc[2] = s.bc[3] = 0.f; s.used[0] = s.used[1] = s.used[2] = s.used[3] = false;
    float sq_dist = 1e18f;
    int degenerate = 0; bool check_simplex = false;
    for (int iter = 0;; ) {
        V3 dir_a = tmul(R, -axis);                                  // (-axis) * basisA
        V3 dir_b = axis;                                            // axis * identity
        V3 p_in_a = v3(dir_a.x >= 0.f ? core.x : -core.x, dir_a.y >= 0.f ? core.y : -core.y, dir_a.z >= 0.f ? core.z : -core.z);
        float d0 = dot(dir_b, tv[0]), d1 = dot(dir_b, tv[1]), d2 = dot(dir_b, tv[2]);
        int mi = d0 < d1 ? (d1 < d2 ? 2 : 1) : (d0 < d2 ? 2 : 0);  // btVector3::maxAxis
        V3 q_in_b = mi == 0 ? tv[0] : (mi == 1 ? tv[1] : tv[2]);
        V3 pw = (R * p_in_a) + oa;
        V3 qw = q_in_b + ob;
        V3 w = pw - qw;
        float delta = dot(axis, w);
        if (delta > 0.f && delta * delta > sq_dist * max_d2) { degenerate = 10; check_simplex = true; break; }
        {   // inSimplex
            bool found = false;
            if (s.n > 0 && v3_eq(s.w[0], w)) found = true;
            if (s.n > 1 && v3_eq(s.w[1], w)) found = true;
            if (s.n > 2 && v3_eq(s.w[2], w)) found = true;
            if (s.n > 3 && v3_eq(s.w[3], w)) found = true;
            if (v3_eq(w, s.last_w)) found = true;
            if (found) { degenerate = 1; check_simplex = true; break; }
        }
        float f0 = sq_dist - delta, f1 = sq_dist * GJK_REL_ERROR2;
        if (f0 <= f1) { degenerate = f0 <= 0.f ? 2 : 11; check_simplex = true; break; }
        s.last_w = w; s.needs_update = true;
        gjk_append(s, w, pw, qw);
#ifdef RLG_GJK_STATS
        RLG_GJK_STATS(4 + s.n, 0);
#endif
        if (!gjk_update(s)) { degenerate = 3; check_simplex = true; break; }
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
    RLG_GJK_STATS(2, s.n * 100 + 0); RLG_GJK_STATS(3, 0);
#endif
    bool valid = false; float distance = 0.f; V3 normal = v3(0, 0, 0), pa = v3(0, 0, 0), pb = v3(0, 0, 0);
    if (check_simplex) {
        gjk_update(s);
        pa = s.cp1; pb = s.cp2;
        normal = axis;
        float l2 = len2(axis);
        if (l2 < 0.0001f) degenerate = 5;
        if (l2 > SIMD_EPS * SIMD_EPS) {
            float rlen = 1.f / sqrtf(l2);
            normal *= rlen;
            float sd = sqrtf(sq_dist);
            pa -= axis * (margin_a / sd);
            pb += axis * (0.f / sd);
            distance = (1.f / rlen) - margin;
            valid = true;
        }
    }
    // btGjkPairDetector.cpp:856-927: penetration (or a degenerate ending with the cores closer than 0.01): the reference asks EPA
    if (!valid || (degenerate && (distance + margin) < 0.01f)) {
        deep = true;
        if (!valid) return false;   // the caller answers with the core polytopes' minimum-translation axis
        // valid but close: EPA's answer replaces GJK's only when it is deeper; both measure the same rounded shapes, so GJK's stands
    }
    if (!(distance < 0.f || distance * distance < max_d2)) return false;
    {   // m_fixContactNormalDirection (:929-948): the normal must point from the triangle's box centre towards the hitbox's
        V3 e = abs_rows_dot(R, v3(core.x + margin_a, core.y + margin_a, core.z + margin_a));
        V3 amin = oa - e, amax = oa + e;
        V3 pos_a = (amax + amin) * 0.5f;
        V3 bmin = vmin(vmin(tv[0] + ob, tv[1] + ob), tv[2] + ob), bmax = vmax(vmax(tv[0] + ob, tv[1] + ob), tv[2] + ob);
        V3 pos_b = (bmin + bmax) * 0.5f;
        if (dot(pos_a - pos_b, normal) < 0.f) normal *= -1.f;
    }
    out.n = normal; out.pb = pb + offset; out.dist = distance;
    return true;
}

}  // namespace rlg
