// arena_epa.h — penetration depth of the hitbox against one mesh triangle (or the ball) where the reference asks for it:
// btGjkPairDetector's "penetration case" (btGjkPairDetector.cpp:847-927) -> btGjkEpaPenetrationDepthSolver::calcPenDepth
// (btGjkEpaPenetrationDepthSolver.cpp:24-79) -> btGjkEpaSolver2::Penetration / Distance (btGjkEpa2.cpp:912-1010), i.e. the SECOND GJK of
// Bullet (gjkepa2_impl::GJK, btGjkEpa2.cpp:158-555: its own simplex code, not btVoronoiSimplexSolver) on the Minkowski difference of the
// margin-inflated shapes expressed in the box's local frame, and EPA on its final simplex (gjkepa2_impl::EPA, :557-872).
//
// Everything is restated operation by operation (fp32, the SSE summation orders of btVector3 / btMatrix3x3 as in rl_math.h) so that the
// contact point a deep hitbox-triangle pair yields is the reference's, bit for bit; oracle/ref_driver.cpp:ref_gjk_box_triangle runs the
// reference's own detector on raw inputs and tools/gjk_fuzz.py + tests/golden/narrowphase_golden.npz compare.
//
// Bullet (zlib licence) copyright notice for the algorithm restated here:
//   Bullet Continuous Collision Detection and Physics Library, Copyright (c) 2003-2008 Erwin Coumans; GJK-EPA collision solver by
//   Nathanael Presson, 2008.  This software is provided 'as-is', without any express or implied warranty; permission is granted to
//   anyone to use it for any purpose and to alter and redistribute it freely, provided its origin is not misrepresented, altered
//   versions are plainly marked as such, and this notice is not removed.  (This file is an altered restatement, not the original.)
//
// Memory: EPA keeps up to 128 support vertices and 256 faces (btGjkEpa2.cpp:58,72).  On the device that state lives in a caller-provided
// ARENA (LDS or global memory, a generic pointer): nothing here is a dynamically indexed local, so nothing goes to scratch memory.  A small
// arena (fewer vertices / faces than Bullet's caps) answers almost every query; when it is too small the routine says so (EPA_ARENA_FULL)
// and the caller repeats the query in a full-size arena, where running out means what it means in Bullet (OutOfVertices / OutOfFaces).
#pragma once
#include "rl_math.h"

#ifndef RLG_EPA_PROF
#define RLG_EPA_PROF(i) ((void)0)   /* phase stamps for tools/probes/epa_probe.hip: 0 start, 1 after the margin GJK, 2 after EncloseOrigin + the first four faces, 3 after the EPA loop, 4 end */
#endif

namespace rlg {

constexpr int   EPA_BT_MAX_VERTICES = 128, EPA_BT_MAX_FACES = 256, EPA_BT_MAX_ITERATIONS = 255, EPA_GJK_MAX_ITERATIONS = 128;
constexpr float EPA_GJK_ACCURACY = 0.0001f, EPA_GJK_MIN_DISTANCE = 0.0001f, EPA_GJK_DUPLICATED_EPS = 0.0001f;
constexpr float EPA_ACCURACY = 0.0001f, EPA_PLANE_EPS = 0.00001f;

struct EpaSV { V3 d, w; };                        // GJK::sSV: direction and Minkowski support point
struct alignas(16) EpaFace {                      // EPA::sFace with indices for pointers, packed so that a face is two 16-byte accesses
    V3 n; float d;
    uint32_t cp;                                  // c[0] | c[1] << 8 | c[2] << 16 | pass << 24: vertex slots (0..3: the simplex GJK ended on, 4..: EPA's own), pass
    uint32_t fe;                                  // f[0] | f[1] << 8 | f[2] << 16 | e[0] << 24 | e[1] << 26 | e[2] << 28: neighbour face / its edge, per edge
    uint32_t links;                               // hull list: (prev + 1) | (next + 1) << 16, 0 = none; the next half doubles as the free list's link
    uint32_t _pad;
};
RLG_HD int epa_face_c(uint32_t cp, int k) { return (int)((cp >> (8 * k)) & 255u); }
RLG_HD int epa_face_pass(uint32_t cp) { return (int)(cp >> 24); }
RLG_HD int epa_face_f(uint32_t fe, int k) { return (int)((fe >> (8 * k)) & 255u); }
RLG_HD int epa_face_e(uint32_t fe, int k) { return (int)((fe >> (24 + 2 * k)) & 3u); }
RLG_HD uint32_t epa_face_set_fe(uint32_t fe, int k, int f, int e) { return (fe & ~((255u << (8 * k)) | (3u << (24 + 2 * k)))) | ((uint32_t)f << (8 * k)) | ((uint32_t)e << (24 + 2 * k)); }
RLG_HD int epa_link_prev(uint32_t l) { return (int)(l & 0xffffu) - 1; }
RLG_HD int epa_link_next(uint32_t l) { return (int)(l >> 16) - 1; }
RLG_HD uint32_t epa_links(int prev, int next) { return (uint32_t)(prev + 1) | ((uint32_t)(next + 1) << 16); }
struct EpaGjkState {                              // GJK's fields and the locals of Evaluate that are indexed at run time
    EpaSV sv[2][4]; float p[2][4]; int rank[2]; int cur;
    V3 lastw[4];
    V3 ray; float distance; int status;           // 0 Valid, 1 Inside, 2 Failed
};
// The two shapes in shape 0's (the box's) local frame: gjkepa2_impl::MinkowskiDiff after Initialize (btGjkEpa2.cpp:874-891) with
// wtrs0 = (R0, o0), wtrs1 = (identity, o1): m_toshape1 = wtrs1.basis^T * wtrs0.basis = R0, m_toshape0 = wtrs0.inverseTimes(wtrs1) =
// (R0^T, (o1 - o0) * R0).  Shape 1 is up to three points (a mesh triangle; the ball's btSphereShape is the single point 0) plus a margin.
struct EpaShapes {
    M3 R0; V3 o0, o1; V3 core; float margin_a;
    V3 t0, t1, t2; float margin_b;
    V3 to0_origin;
};
enum { EPA_VALID = 0, EPA_TOUCHING, EPA_DEGENERATED, EPA_NONCONVEX, EPA_INVALIDHULL, EPA_OUTOFFACES, EPA_OUTOFVERTICES, EPA_ACCURACY_REACHED, EPA_FALLBACK, EPA_FAILED,
       EPA_ARENA_FULL = 100 };
struct EpaRun {
    int status;
    int hull_root, hull_count;      // m_hull
    int free_root, next_fresh;      // m_stock = freed faces (LIFO) followed by the never-used ones in index order (Initialize, :637-647)
    int nextsv;
    bool arena_full;
};
struct EpaResult { int status; V3 normal; float depth; int rank; EpaSV c[3]; float p[3]; };
struct EpaArena {
    EpaShapes* sh;                                // the pair, in the box's frame (kept here so that the out-of-line routines below share it without a stack copy)
    EpaGjkState* g;
    EpaRun* run; EpaResult* res;    // EPA's fields / m_result (kept in the arena: the routines below are real calls on the device)
    EpaSV* sv;                                    // [4 + cap_v]
    EpaFace* fc;                                  // [cap_f]
    uint16_t* stack;                              // [cap_f]: the recursion of EPA::expand
    int cap_v, cap_f;
};

// Address-space hints (device builds): RLG_EPA_IN_LDS(x) tells the optimiser that x lives in LDS -- ds_read / ds_write instead of flat
// accesses -- in the instantiations that only ever see the LDS arenas.
#ifndef RLG_EPA_IN_LDS
#define RLG_EPA_IN_LDS(ref) ((void)0)
#endif
#define RLG_EPA_ARENA_IN_LDS(A) do { RLG_EPA_IN_LDS(*(A).sh); RLG_EPA_IN_LDS(*(A).g); RLG_EPA_IN_LDS(*(A).run); RLG_EPA_IN_LDS(*(A).res); RLG_EPA_IN_LDS(*(A).sv); RLG_EPA_IN_LDS(*(A).fc); RLG_EPA_IN_LDS(*(A).stack); } while (0)
constexpr size_t epa_arena_bytes(int cap_v, int cap_f) {
    return sizeof(EpaShapes) + sizeof(EpaGjkState) + sizeof(EpaRun) + sizeof(EpaResult) + sizeof(EpaSV) * (size_t)(4 + cap_v) + sizeof(EpaFace) * (size_t)cap_f + 2 * (size_t)cap_f + 16;
}
RLG_HD EpaArena epa_arena_at(void* mem, int cap_v, int cap_f) {
    EpaArena a; unsigned char* p = reinterpret_cast<unsigned char*>(mem);
    a.sh = reinterpret_cast<EpaShapes*>(p); p += sizeof(EpaShapes);
    a.g = reinterpret_cast<EpaGjkState*>(p); p += sizeof(EpaGjkState);
    a.run = reinterpret_cast<EpaRun*>(p); p += sizeof(EpaRun);
    a.res = reinterpret_cast<EpaResult*>(p); p += sizeof(EpaResult);
    a.sv = reinterpret_cast<EpaSV*>(p); p += sizeof(EpaSV) * (size_t)(4 + cap_v);
    p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 15u) & ~(uintptr_t)15u);
    a.fc = reinterpret_cast<EpaFace*>(p); p += sizeof(EpaFace) * (size_t)cap_f;
    a.stack = reinterpret_cast<uint16_t*>(p);
    a.cap_v = cap_v; a.cap_f = cap_f;
    return a;
}

RLG_HD EpaShapes epa_shapes(const M3& R0, V3 o0, V3 core, float margin_a, V3 o1, V3 t0, V3 t1, V3 t2, float margin_b) {
    EpaShapes s; s.R0 = R0; s.o0 = o0; s.o1 = o1; s.core = core; s.margin_a = margin_a; s.t0 = t0; s.t1 = t1; s.t2 = t2; s.margin_b = margin_b;
    s.to0_origin = tmul(R0, o1 - o0);             // v * m_basis (btTransform.h:218-223)
    return s;
}
// btConvexShape::localGetSupportVertexNonVirtual's direction (btConvexShape.cpp:183-193)
RLG_HD V3 epa_dir_norm(V3 d) {
    if (len2(d) < SIMD_EPS * SIMD_EPS) d = v3(-1.f, -1.f, -1.f);
    return normalized(d);
}
RLG_HD V3 epa_box_vertex(const EpaShapes& s, V3 d) {   // btBoxShape: btFsels(d, h, -h) per axis (btConvexShape.cpp:134-151)
    return v3(d.x >= 0.f ? s.core.x : -s.core.x, d.y >= 0.f ? s.core.y : -s.core.y, d.z >= 0.f ? s.core.z : -s.core.z);
}
RLG_HD V3 epa_b_vertex(const EpaShapes& s, V3 d) {     // btTriangleShape: dots.maxAxis() (btConvexShape.cpp:152-160); one point for the sphere
    const float d0 = dot(d, s.t0), d1 = dot(d, s.t1), d2 = dot(d, s.t2);
    const int mi = d0 < d1 ? (d1 < d2 ? 2 : 1) : (d0 < d2 ? 2 : 0);
    return v3(mi == 0 ? s.t0.x : (mi == 1 ? s.t1.x : s.t2.x), mi == 0 ? s.t0.y : (mi == 1 ? s.t1.y : s.t2.y), mi == 0 ? s.t0.z : (mi == 1 ? s.t1.z : s.t2.z));
}
RLG_HD V3 epa_support0(const EpaShapes& s, V3 d, bool margins) {
    if (!margins) return epa_box_vertex(s, d);
    const V3 n = epa_dir_norm(d);
    return epa_box_vertex(s, n) + n * s.margin_a;
}
RLG_HD V3 epa_support1(const EpaShapes& s, V3 d, bool margins) {   // m_toshape0 * Ls(m_toshape1 * d)
    V3 dl = s.R0 * d;
    V3 p;
    if (margins) { const V3 n = epa_dir_norm(dl); p = epa_b_vertex(s, n) + n * s.margin_b; }
    else p = epa_b_vertex(s, dl);
    return tmul(s.R0, p) + s.to0_origin;
}
RLG_HD V3 epa_support(const EpaShapes& s, V3 d, bool margins) { return epa_support0(s, d, margins) - epa_support1(s, -d, margins); }
RLG_HD_NOINLINE void epa_getsupport(const EpaShapes& s, bool margins, V3 d, EpaSV& sv) {   // GJK::getsupport (:422-426)
    RLG_EPA_IN_LDS(s); RLG_EPA_IN_LDS(sv);
    sv.d = vdiv_bt(d, len(d));
    sv.w = epa_support(s, sv.d, margins);
}
RLG_HD float epa_det(V3 a, V3 b, V3 c) {   // GJK::det (:437-442)
    return (a.y * b.z * c.x + a.z * b.x * c.y - a.x * b.z * c.y - a.y * b.x * c.z + a.x * b.y * c.z - a.z * b.y * c.x);
}

// GJK::projectorigin, 2 / 3 / 4 points (:443-554).  w / m are only written where the reference writes them.
RLG_HD float epa_project2(V3 a, V3 b, float& w0, float& w1, uint32_t& m) {
    const V3 d = b - a;
    const float l = len2(d);
    if (l > 0.f) {
        const float t = l > 0.f ? -dot(a, d) / l : 0.f;
        if (t >= 1.f) { w0 = 0.f; w1 = 1.f; m = 2u; return len2(b); }
        else if (t <= 0.f) { w0 = 1.f; w1 = 0.f; m = 1u; return len2(a); }
        else { w1 = t; w0 = 1.f - t; m = 3u; return len2(a + d * t); }
    }
    return -1.f;
}
struct EpaW3 { float w0, w1, w2; };
RLG_HD void epa_w3_set(EpaW3& w, int i, float v) { if (i == 0) w.w0 = v; else if (i == 1) w.w1 = v; else w.w2 = v; }
RLG_HD_NOINLINE float epa_project3(V3 a, V3 b, V3 c, EpaW3& w, uint32_t& m) {
    const V3 dl0 = a - b, dl1 = b - c, dl2 = c - a;
    const V3 n = cross(dl0, dl1);
    const float l = len2(n);
    if (l > 0.f) {
        float mindist = -1.f;
        float subw0 = 0.f, subw1 = 0.f; uint32_t subm = 0u;
        RLG_UNROLL
        for (int i = 0; i < 3; i++) {
            const int j = i == 2 ? 0 : i + 1, k = j == 2 ? 0 : j + 1;   // imd3
            const V3 vi = i == 0 ? a : (i == 1 ? b : c), vj = j == 0 ? a : (j == 1 ? b : c);
            const V3 dli = i == 0 ? dl0 : (i == 1 ? dl1 : dl2);
            if (dot(vi, cross(dli, n)) > 0.f) {
                const float subd = epa_project2(vi, vj, subw0, subw1, subm);
                if ((mindist < 0.f) || (subd < mindist)) {
                    mindist = subd;
                    m = ((subm & 1u) ? 1u << i : 0u) + ((subm & 2u) ? 1u << j : 0u);
                    epa_w3_set(w, i, subw0); epa_w3_set(w, j, subw1); epa_w3_set(w, k, 0.f);
                }
            }
        }
        if (mindist < 0.f) {
            const float d = dot(a, n);
            const float s = sqrtf(l);
            const V3 p = n * (d / l);
            mindist = len2(p);
            m = 7u;
            w.w0 = len(cross(dl1, b - p)) / s;
            w.w1 = len(cross(dl2, c - p)) / s;
            w.w2 = 1.f - (w.w0 + w.w1);
        }
        return mindist;
    }
    return -1.f;
}
struct EpaW4 { float w0, w1, w2, w3; };
RLG_HD void epa_w4_set(EpaW4& w, int i, float v) { if (i == 0) w.w0 = v; else if (i == 1) w.w1 = v; else if (i == 2) w.w2 = v; else w.w3 = v; }
RLG_HD_NOINLINE float epa_project4(V3 a, V3 b, V3 c, V3 d, EpaW4& w, uint32_t& m) {
    const V3 dl0 = a - d, dl1 = b - d, dl2 = c - d;
    const float vl = epa_det(dl0, dl1, dl2);
    const bool ng = (vl * dot(a, cross(b - c, a - b))) <= 0.f;
    if (ng && (fabsf(vl) > 0.f)) {
        float mindist = -1.f;
        EpaW3 subw; subw.w0 = subw.w1 = subw.w2 = 0.f; uint32_t subm = 0u;
        RLG_UNROLL
        for (int i = 0; i < 3; i++) {
            const int j = i == 2 ? 0 : i + 1, k = j == 2 ? 0 : j + 1;
            const V3 vi = i == 0 ? a : (i == 1 ? b : c), vj = j == 0 ? a : (j == 1 ? b : c);
            const V3 dli = i == 0 ? dl0 : (i == 1 ? dl1 : dl2), dlj = j == 0 ? dl0 : (j == 1 ? dl1 : dl2);
            const float s = vl * dot(d, cross(dli, dlj));
            if (s > 0.f) {
                const float subd = epa_project3(vi, vj, d, subw, subm);
                if ((mindist < 0.f) || (subd < mindist)) {
                    mindist = subd;
                    m = ((subm & 1u) ? 1u << i : 0u) + ((subm & 2u) ? 1u << j : 0u) + ((subm & 4u) ? 8u : 0u);
                    epa_w4_set(w, i, subw.w0); epa_w4_set(w, j, subw.w1); epa_w4_set(w, k, 0.f); w.w3 = subw.w2;
                }
            }
        }
        if (mindist < 0.f) {
            mindist = 0.f;
            m = 15u;
            w.w0 = epa_det(c, b, d) / vl;
            w.w1 = epa_det(a, c, d) / vl;
            w.w2 = epa_det(b, a, d) / vl;
            w.w3 = 1.f - (w.w0 + w.w1 + w.w2);
        }
        return mindist;
    }
    return -1.f;
}

// GJK::Evaluate (:204-347).  The simplices hold their vertices by value (the reference's pointer / free-list bookkeeping only shares them).
RLG_HD_NOINLINE int epa_gjk_evaluate(EpaGjkState& G, const EpaShapes& sh, bool margins, V3 guess) {
    RLG_EPA_IN_LDS(G); RLG_EPA_IN_LDS(sh);
    int iterations = 0;
    float sqdist = 0.f, alpha = 0.f;
    int clastw = 0;
    G.cur = 0; G.status = 0; G.distance = 0.f;
    G.rank[0] = 0; G.rank[1] = 0;
    G.ray = guess;
    const float sqrl = len2(G.ray);
    epa_getsupport(sh, margins, sqrl > 0.f ? -G.ray : v3(1.f, 0.f, 0.f), G.sv[0][0]); G.p[0][0] = 0.f; G.rank[0] = 1;
    G.p[0][0] = 1.f;
    G.ray = G.sv[0][0].w;
    sqdist = sqrl;
    G.lastw[0] = G.lastw[1] = G.lastw[2] = G.lastw[3] = G.ray;
    do {
        const int cur = G.cur, next = 1 - cur;
        const float rl = len(G.ray);
        if (rl < EPA_GJK_MIN_DISTANCE) { G.status = 1; break; }
        {   // appendvertice(cs, -m_ray)
            const int r = G.rank[cur];
            G.p[cur][r] = 0.f;
            epa_getsupport(sh, margins, -G.ray, G.sv[cur][r]);
            G.rank[cur] = r + 1;
        }
        const V3 w = G.sv[cur][G.rank[cur] - 1].w;
        bool found = false;
        for (int i = 0; i < 4; ++i) if (len2(w - G.lastw[i]) < EPA_GJK_DUPLICATED_EPS) { found = true; break; }
        if (found) { G.rank[cur]--; break; }
        else { clastw = (clastw + 1) & 3; G.lastw[clastw] = w; }
        const float omega = dot(G.ray, w) / rl;
        alpha = omega > alpha ? omega : alpha;        // btMax(omega, alpha)
        if (((rl - alpha) - (EPA_GJK_ACCURACY * rl)) <= 0.f) { G.rank[cur]--; break; }
        EpaW4 wt; wt.w0 = wt.w1 = wt.w2 = wt.w3 = 0.f;   // (the reference's weights[] are uninitialised; only entries with a mask bit are read)
        uint32_t mask = 0u;
        const int rk = G.rank[cur];
        if (rk == 2) sqdist = epa_project2(G.sv[cur][0].w, G.sv[cur][1].w, wt.w0, wt.w1, mask);
        else if (rk == 3) { EpaW3 w3; w3.w0 = w3.w1 = w3.w2 = 0.f; sqdist = epa_project3(G.sv[cur][0].w, G.sv[cur][1].w, G.sv[cur][2].w, w3, mask); wt.w0 = w3.w0; wt.w1 = w3.w1; wt.w2 = w3.w2; }
        else if (rk == 4) sqdist = epa_project4(G.sv[cur][0].w, G.sv[cur][1].w, G.sv[cur][2].w, G.sv[cur][3].w, wt, mask);
        if (sqdist >= 0.f) {
            G.rank[next] = 0;
            G.ray = v3(0.f, 0.f, 0.f);
            G.cur = next;
            for (int i = 0; i < rk; ++i) {
                if (mask & (1u << i)) {
                    const float wi = i == 0 ? wt.w0 : (i == 1 ? wt.w1 : (i == 2 ? wt.w2 : wt.w3));
                    const int nr = G.rank[next];
                    G.sv[next][nr] = G.sv[cur][i];
                    G.p[next][nr] = wi;
                    G.rank[next] = nr + 1;
                    G.ray += G.sv[cur][i].w * wi;
                }
            }
            if (mask == 15u) G.status = 1;
        } else { G.rank[cur]--; break; }
        G.status = ((++iterations) < EPA_GJK_MAX_ITERATIONS) ? G.status : 2;
    } while (G.status == 0);
    if (G.status == 0) G.distance = len(G.ray);
    else if (G.status == 1) G.distance = 0.f;
    return G.status;
}

// GJK::EncloseOrigin (:348-420) for the ranks EPA::Evaluate can enter it with (>= 2); the recursion is three nested levels.
RLG_HD void epa_gjk_push(EpaGjkState& G, const EpaShapes& sh, bool margins, V3 v) {
    const int c = G.cur, r = G.rank[c];
    G.p[c][r] = 0.f; epa_getsupport(sh, margins, v, G.sv[c][r]); G.rank[c] = r + 1;
}
RLG_HD bool epa_enclose4(EpaGjkState& G) {
    const EpaSV* s = G.sv[G.cur];
    return fabsf(epa_det(s[0].w - s[3].w, s[1].w - s[3].w, s[2].w - s[3].w)) > 0.f;
}
RLG_HD bool epa_enclose3(EpaGjkState& G, const EpaShapes& sh, bool margins) {
    const EpaSV* s = G.sv[G.cur];
    const V3 n = cross(s[1].w - s[0].w, s[2].w - s[0].w);
    if (len2(n) > 0.f) {
        epa_gjk_push(G, sh, margins, n);
        if (epa_enclose4(G)) return true;
        G.rank[G.cur]--;
        epa_gjk_push(G, sh, margins, -n);
        if (epa_enclose4(G)) return true;
        G.rank[G.cur]--;
    }
    return false;
}
RLG_HD bool epa_enclose2(EpaGjkState& G, const EpaShapes& sh, bool margins) {
    const V3 d = G.sv[G.cur][1].w - G.sv[G.cur][0].w;
    for (int i = 0; i < 3; ++i) {
        const V3 axis = v3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f);
        const V3 p = cross(d, axis);
        if (len2(p) > 0.f) {
            epa_gjk_push(G, sh, margins, p);
            if (epa_enclose3(G, sh, margins)) return true;
            G.rank[G.cur]--;
            epa_gjk_push(G, sh, margins, -p);
            if (epa_enclose3(G, sh, margins)) return true;
            G.rank[G.cur]--;
        }
    }
    return false;
}
RLG_HD_NOINLINE bool epa_enclose_origin(EpaGjkState& G, const EpaShapes& sh, bool margins) {
    RLG_EPA_IN_LDS(G); RLG_EPA_IN_LDS(sh);
    const int r = G.rank[G.cur];
    if (r == 2) return epa_enclose2(G, sh, margins);
    if (r == 3) return epa_enclose3(G, sh, margins);
    if (r == 4) return epa_enclose4(G);
    return false;
}

// ---- EPA (:557-872) ---------------------------------------------------------------------------------------------------------------------
RLG_HD void epa_bind(const EpaArena& A, int fa, int ea, int fb, int eb) {
    A.fc[fa].fe = epa_face_set_fe(A.fc[fa].fe, ea, fb, eb);
    A.fc[fb].fe = epa_face_set_fe(A.fc[fb].fe, eb, fa, ea);
}
RLG_HD void epa_hull_append(const EpaArena& A, int f) {
    EpaRun& E = *A.run;
    const int root = E.hull_root;
    A.fc[f].links = epa_links(-1, root);
    if (root >= 0) A.fc[root].links = epa_links(f, epa_link_next(A.fc[root].links));
    E.hull_root = f; E.hull_count++;
}
RLG_HD void epa_hull_remove(const EpaArena& A, int f) {
    EpaRun& E = *A.run;
    const uint32_t l = A.fc[f].links; const int prev = epa_link_prev(l), next = epa_link_next(l);
    if (next >= 0) A.fc[next].links = epa_links(prev, epa_link_next(A.fc[next].links));
    if (prev >= 0) A.fc[prev].links = epa_links(epa_link_prev(A.fc[prev].links), next);
    if (f == E.hull_root) E.hull_root = next;
    E.hull_count--;
}
RLG_HD void epa_stock_push(const EpaArena& A, int f) { A.fc[f].links = epa_links(-1, A.run->free_root); A.run->free_root = f; }
RLG_HD bool epa_edge_dist(V3 fn, V3 aw, V3 bw, float& dist) {   // EPA::getedgedist (:743-779)
    const V3 ba = bw - aw;
    const V3 n_ab = cross(ba, fn);
    const float a_dot_nab = dot(aw, n_ab);
    if (a_dot_nab < 0.f) {
        const float ba_l2 = len2(ba);
        const float a_dot_ba = dot(aw, ba);
        const float b_dot_ba = dot(bw, ba);
        if (a_dot_ba > 0.f) dist = len(aw);
        else if (b_dot_ba < 0.f) dist = len(bw);
        else {
            const float a_dot_b = dot(aw, bw);
            const float q = (len2(aw) * len2(bw) - a_dot_b * a_dot_b) / ba_l2;
            dist = sqrtf(q > 0.f ? q : 0.f);           // btMax(q, 0)
        }
        return true;
    }
    return false;
}
// EPA::newface (:780-824); -1 = none (m_status says why)
RLG_HD_NOINLINE int epa_newface(EpaArena A, int a, int b, int c, bool forced) {
    RLG_EPA_ARENA_IN_LDS(A);
    EpaRun& E = *A.run;
    int face;
    if (E.free_root >= 0) { face = E.free_root; E.free_root = epa_link_next(A.fc[face].links); }
    else if (E.next_fresh < A.cap_f) face = E.next_fresh++;
    else {
        if (A.cap_f < EPA_BT_MAX_FACES) { E.arena_full = true; return -1; }
        E.status = EPA_OUTOFFACES;                    // m_stock.root == 0
        return -1;
    }
    epa_hull_append(A, face);
    EpaFace& F = A.fc[face];
    const V3 aw = A.sv[a].w, bw = A.sv[b].w, cw = A.sv[c].w;
    F.cp = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16);   // pass = 0
    V3 n = cross(bw - aw, cw - aw);
    const float l = len(n);
    const bool v = l > EPA_ACCURACY;
    if (v) {
        float d;
        if (!(epa_edge_dist(n, aw, bw, d) || epa_edge_dist(n, bw, cw, d) || epa_edge_dist(n, cw, aw, d))) d = dot(aw, n) / l;
        F.d = d;
        F.n = vdiv_bt(n, l);
        if (forced || (d >= -EPA_PLANE_EPS)) return face;
        else E.status = EPA_NONCONVEX;
    } else { F.n = n; E.status = EPA_DEGENERATED; }
    epa_hull_remove(A, face);
    epa_stock_push(A, face);
    return -1;
}
RLG_HD int epa_findbest(const EpaArena& A) {   // EPA::findbest (:825-839)
    int minf = A.run->hull_root;
    float mind = A.fc[minf].d * A.fc[minf].d;
    for (int f = epa_link_next(A.fc[minf].links); f >= 0; f = epa_link_next(A.fc[f].links)) {
        const float sqd = A.fc[f].d * A.fc[f].d;
        if (sqd < mind) { minf = f; mind = sqd; }
    }
    return minf;
}
struct EpaHorizon { int cf, ff, nf; };
// EPA::expand (:840-871), its recursion unrolled onto A.stack: entry = face | edge << 8 | stage << 10
RLG_HD_NOINLINE bool epa_expand(EpaArena A, int pass, int w, int f0, int e0, EpaHorizon& hz) {
    RLG_EPA_ARENA_IN_LDS(A);
    EpaRun& E = *A.run;
    uint16_t* st = A.stack; int sp = 0;
    st[sp++] = (uint16_t)(f0 | (e0 << 8));
    bool ret = false;
    const V3 ww = A.sv[w].w;
    while (sp > 0) {
        const uint16_t top = st[sp - 1];
        const int f = top & 0xff, e = (top >> 8) & 3, stage = top >> 10;
        EpaFace& F = A.fc[f];
        const int e1 = e == 2 ? 0 : e + 1, e2 = e == 0 ? 2 : e - 1;   // i1m3, i2m3
        if (stage == 0) {
            const uint32_t cp = F.cp;
            if (epa_face_pass(cp) != (pass & 255)) {
                if ((dot(F.n, ww) - F.d) < -EPA_PLANE_EPS) {
                    const int nf = epa_newface(A, epa_face_c(cp, e1), epa_face_c(cp, e), w, false);
                    if (nf >= 0) {
                        epa_bind(A, nf, 0, f, e);
                        if (hz.cf >= 0) epa_bind(A, hz.cf, 1, nf, 2); else hz.ff = nf;
                        hz.cf = nf; ++hz.nf;
                        ret = true;
                    } else ret = false;
                    sp--;
                } else {
                    F.cp = (cp & 0x00ffffffu) | ((uint32_t)(pass & 255) << 24);
                    st[sp - 1] = (uint16_t)(f | (e << 8) | (1 << 10));
                    if (sp >= A.cap_f) { E.arena_full = true; return false; }
                    const uint32_t fe = F.fe;
                    st[sp++] = (uint16_t)(epa_face_f(fe, e1) | (epa_face_e(fe, e1) << 8));
                }
            } else { ret = false; sp--; }
        } else if (stage == 1) {
            if (ret) {
                st[sp - 1] = (uint16_t)(f | (e << 8) | (2 << 10));
                if (sp >= A.cap_f) { E.arena_full = true; return false; }
                const uint32_t fe = F.fe;
                st[sp++] = (uint16_t)(epa_face_f(fe, e2) | (epa_face_e(fe, e2) << 8));
            } else { ret = false; sp--; }
        } else {
            if (ret) { epa_hull_remove(A, f); epa_stock_push(A, f); ret = true; }
            sp--;
        }
        if (E.arena_full) return false;
    }
    return ret;
}

// EPA::Evaluate (:648-742); the result goes to *A.res
RLG_HD_NOINLINE int epa_evaluate(EpaArena A, bool margins, V3 guess) {
    RLG_EPA_ARENA_IN_LDS(A);
    EpaGjkState& G = *A.g; const EpaShapes& sh = *A.sh; EpaResult& out = *A.res;
    EpaRun& E = *A.run; E.status = EPA_FAILED; E.hull_root = -1; E.hull_count = 0; E.free_root = -1; E.next_fresh = 0; E.nextsv = 0; E.arena_full = false;
    if ((G.rank[G.cur] > 1) && epa_enclose_origin(G, sh, margins)) {
        E.status = EPA_VALID;
        for (int i = 0; i < 4; i++) A.sv[i] = G.sv[G.cur][i];
        if (epa_det(A.sv[0].w - A.sv[3].w, A.sv[1].w - A.sv[3].w, A.sv[2].w - A.sv[3].w) < 0.f) {
            const EpaSV t = A.sv[0]; A.sv[0] = A.sv[1]; A.sv[1] = t;   // btSwap(simplex.c[0], simplex.c[1]) (the weights p are not read again)
        }
        const int t0 = epa_newface(A, 0, 1, 2, true);
        const int t1 = epa_newface(A, 1, 0, 3, true);
        const int t2 = epa_newface(A, 2, 1, 3, true);
        const int t3 = epa_newface(A, 0, 2, 3, true);
        if (E.arena_full) return EPA_ARENA_FULL;
        RLG_EPA_PROF(2);
        if (E.hull_count == 4) {
            int best = epa_findbest(A);
            EpaFace outer = A.fc[best];
            int pass = 0, iterations = 0;
            epa_bind(A, t0, 0, t1, 0);
            epa_bind(A, t0, 1, t2, 0);
            epa_bind(A, t0, 2, t3, 0);
            epa_bind(A, t1, 1, t3, 2);
            epa_bind(A, t1, 2, t2, 1);
            epa_bind(A, t2, 2, t3, 1);
            E.status = EPA_VALID;
            for (; iterations < EPA_BT_MAX_ITERATIONS; ++iterations) {
                if (E.nextsv < EPA_BT_MAX_VERTICES) {
                    if (E.nextsv >= A.cap_v) return EPA_ARENA_FULL;
                    EpaHorizon hz; hz.cf = -1; hz.ff = -1; hz.nf = 0;
                    const int w = 4 + E.nextsv++;
                    bool valid = true;
                    ++pass;
                    A.fc[best].cp = (A.fc[best].cp & 0x00ffffffu) | ((uint32_t)(pass & 255) << 24);
                    const V3 bn = A.fc[best].n;
                    epa_getsupport(sh, margins, bn, A.sv[w]);
                    const float wdist = dot(bn, A.sv[w].w) - A.fc[best].d;
                    if (wdist > EPA_ACCURACY) {
                        for (int j = 0; (j < 3) && valid; ++j) {
                            const uint32_t bfe = A.fc[best].fe;
                            valid &= epa_expand(A, pass, w, epa_face_f(bfe, j), epa_face_e(bfe, j), hz);
                            if (E.arena_full) return EPA_ARENA_FULL;
                        }
                        if (valid && (hz.nf >= 3)) {
                            epa_bind(A, hz.cf, 1, hz.ff, 2);
                            epa_hull_remove(A, best);
                            epa_stock_push(A, best);
                            best = epa_findbest(A);
                            outer = A.fc[best];
                        } else { E.status = EPA_INVALIDHULL; break; }
                    } else { E.status = EPA_ACCURACY_REACHED; break; }
                } else { E.status = EPA_OUTOFVERTICES; break; }
            }
#ifdef RLG_EPA_STATS
            RLG_EPA_STATS(E.nextsv, E.next_fresh, iterations, E.status);
#endif
            const V3 projection = outer.n * outer.d;
            out.normal = outer.n;
            out.depth = outer.d;
            out.rank = 3;
            out.c[0] = A.sv[epa_face_c(outer.cp, 0)]; out.c[1] = A.sv[epa_face_c(outer.cp, 1)]; out.c[2] = A.sv[epa_face_c(outer.cp, 2)];
            const V3 c0 = out.c[0].w, c1 = out.c[1].w, c2 = out.c[2].w;
            float p0 = len(cross(c1 - projection, c2 - projection));
            float p1 = len(cross(c2 - projection, c0 - projection));
            float p2 = len(cross(c0 - projection, c1 - projection));
            const float sum = p0 + p1 + p2;
            out.p[0] = p0 / sum; out.p[1] = p1 / sum; out.p[2] = p2 / sum;
            out.status = E.status;
            return E.status;
        }
    }
    if (E.arena_full) return EPA_ARENA_FULL;
    // Fallback
    out.status = EPA_FALLBACK;
    V3 nrm = -guess;
    const float nl = len(nrm);
    if (nl > 0.f) nrm = vdiv_bt(nrm, nl); else nrm = v3(1.f, 0.f, 0.f);
    out.normal = nrm;
    out.depth = 0.f;
    out.rank = 1;
    out.c[0] = G.sv[G.cur][0];
    out.p[0] = 1.f;
    return EPA_FALLBACK;
}

// btGjkEpaSolver2::Penetration (:959-1010, usemargins = true) and ::Distance (:912-956) on the pair, then
// btGjkEpaPenetrationDepthSolver::calcPenDepth's loop over its nine guess directions (btGjkEpaPenetrationDepthSolver.cpp:24-79).
// Returns 1: penetration (isValid2 = true), 0: calcPenDepth returned false (`v` and the witnesses may still be set: the second GJK's
// distance), EPA_ARENA_FULL: repeat in a bigger arena.
struct PenDepth { V3 v, wa, wb; };
RLG_HD int epa_calc_pen_depth(EpaArena A, const EpaShapes& shapes, PenDepth& out) {
    RLG_EPA_ARENA_IN_LDS(A);
    *A.sh = shapes;
    const EpaShapes& sh = *A.sh;
    EpaGjkState& G = *A.g;
    for (int i = 0; i < 9; i++) {
        V3 guess;
        if (i == 0) guess = safe_normalized(sh.o1 - sh.o0);
        else if (i == 1) guess = safe_normalized(sh.o0 - sh.o1);
        else guess = v3((i == 4 || i == 5 || i == 6 || i == 8) ? 1.f : 0.f, (i == 3 || i == 5 || i == 6 || i == 7) ? 1.f : 0.f, (i == 2 || i == 6 || i == 7 || i == 8) ? 1.f : 0.f);
        // Penetration
        RLG_EPA_PROF(0);
        const int gs = epa_gjk_evaluate(G, sh, true, -guess);
        RLG_EPA_PROF(1);
        if (gs == 1) {
            const int es = epa_evaluate(A, true, -guess);
            RLG_EPA_PROF(3);
            if (es == EPA_ARENA_FULL) return EPA_ARENA_FULL;
            const EpaResult& r = *A.res;
            V3 w0 = v3(0.f, 0.f, 0.f);
            for (int k = 0; k < r.rank; ++k) w0 += epa_support0(sh, r.c[k].d, true) * r.p[k];
            out.wa = (sh.R0 * w0) + sh.o0;
            out.wb = (sh.R0 * (w0 - r.normal * r.depth)) + sh.o0;
            out.v = -r.normal;
            RLG_EPA_PROF(4);
            return 1;
        }
        // Distance (no margins)
        const int ds = epa_gjk_evaluate(G, sh, false, guess);
        if (ds == 0) {
            V3 w0 = v3(0.f, 0.f, 0.f), w1 = v3(0.f, 0.f, 0.f);
            const int c = G.cur;
            for (int k = 0; k < G.rank[c]; ++k) {
                const float p = G.p[c][k];
                w0 += epa_support0(sh, G.sv[c][k].d, false) * p;
                w1 += epa_support1(sh, -G.sv[c][k].d, false) * p;
            }
            out.wa = (sh.R0 * w0) + sh.o0;
            out.wb = (sh.R0 * w1) + sh.o0;
            V3 n = w0 - w1;
            const float dist = len(n);
            n = vdiv_bt(n, dist > EPA_GJK_MIN_DISTANCE ? dist : 1.f);
            out.v = n;
            return 0;
        }
    }
    out.wa = out.wb = out.v = v3(0.f, 0.f, 0.f);
    return 0;
}

}  // namespace rlg
