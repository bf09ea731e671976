// arena_step.h — one physics tick of one arena: Arena::Step (RocketSim/src/Sim/Arena/Arena.cpp:716-812) with
// btDiscreteDynamicsWorld::stepSimulation (btDiscreteDynamicsWorld.cpp:325-437) restated for the fixed body
// set {ball, NC cars} and the contact list of arena_world.h.
#pragma once
#include "arena_car.h"
#include "arena_gjk.h"

#ifndef RLG_DBG_COUNT
#define RLG_DBG_COUNT(i) ((void)0)
#endif
namespace rlg {

// ---- contact-added callbacks (Arena.cpp:283-427) ----------------------------------------------------------
template <int NC>
RLG_HD void on_car_ball_contact(Arena<NC>& A, int ci, V3 point_rel_ball) {   // contact point on the ball minus the ball centre (BT)
    Car& car = A.cars[ci];
    Ball& ball = A.ball;
    car.flags |= CF_BALLHIT_VALID;
    car.bh_rel_pos = point_rel_ball * BT2UU;  // m_localPoint on the ball (ball basis = identity)
    car.bh_tick_hit = A.tick_count;
    car.bh_ball_pos = ball.b.pos * BT2UU;
    car.bh_extra_hit_vel = v3(0, 0, 0);
    // unsigned arithmetic in the reference: tickCountWhenExtraImpulseApplied starts at ~0ULL ("never") -> +1 wraps to 0
    bool never = car.bh_tick_extra < 0;
    if (never || (A.tick_count > car.bh_tick_extra + 1) || (car.bh_tick_extra > A.tick_count)) car.bh_tick_extra = A.tick_count;
    else return;
    V3 car_fwd = col0(car.b.rot);
    V3 rel_pos = (ball.b.pos * BT2UU) - (car.b.pos * BT2UU);
    V3 rel_vel = (ball.b.vel * BT2UU) - (car.b.vel * BT2UU);
    float rel_speed = fminf(len(rel_vel), K::BALL_CAR_EXTRA_IMPULSE_MAXDELTAVEL_UU);
    if (rel_speed > 0.f) {
        V3 hit_dir = safe_normalized(rel_pos * v3(1, 1, K::BALL_CAR_EXTRA_IMPULSE_Z_SCALE));
        V3 adj = car_fwd * dot(hit_dir, car_fwd) * (1 - K::BALL_CAR_EXTRA_IMPULSE_FORWARD_SCALE);
        hit_dir = safe_normalized(hit_dir - adj);
        V3 added = (hit_dir * rel_speed) * curve_ball_car_extra(rel_speed) * A.mut.ball_hit_extra_scale;
        car.bh_extra_hit_vel = added;
        ball.vel_impulse_cache += added * UU2BT;
    }
}

// Arena::_BtCallback_OnCarCarCollision (Arena.cpp:336-418). local points are in each car's body frame (BT).
template <int NC>
RLG_HD_COLD void on_car_car_contact(Arena<NC>& A, int ia, int ib, V3 local_a, V3 local_b, TickEvents& ev) {
    for (int i = 0; i < 2; i++) {
        bool swapped = (i == 1);
        int i1 = swapped ? ib : ia, i2 = swapped ? ia : ib;
        Car& c1 = A.cars[i1]; Car& c2 = A.cars[i2];
        if ((c1.flags & CF_IS_DEMOED) || (c2.flags & CF_IS_DEMOED)) return;
        if (c1.car_contact_other == i2 + 1 && c1.car_contact_cooldown > 0) continue;
        V3 p1 = c1.b.pos * BT2UU, p2 = c2.b.pos * BT2UU, v1 = c1.b.vel * BT2UU, v2 = c2.b.vel * BT2UU;
        V3 delta = p2 - p1;
        if (dot(v1, delta) > 0) {
            float l1 = len(v1); V3 vel_dir = (l1 > SIMD_EPS * SIMD_EPS) ? vdiv_rs(v1, l1) : v3(0, 0, 0);
            float l2 = len(delta); V3 dir_to = (l2 > SIMD_EPS * SIMD_EPS) ? vdiv_rs(delta, l2) : v3(0, 0, 0);
            float speed_towards = dot(v1, dir_to);
            float other_away = dot(v2, vel_dir);
            if (speed_towards > other_away) {
                V3 lp = swapped ? local_b : local_a;
                bool bumper = (lp.x * BT2UU) > K::BUMP_MIN_FORWARD_DIST;
                if (bumper) {
                    const uint32_t mf = A.mut.flags;   // MutatorConfig::demoMode, enableTeamDemos (Arena.cpp:375-388)
                    bool is_demo = (mf & MUT_DEMO_ON_CONTACT) ? true : (mf & MUT_DEMO_DISABLED) ? false : (c1.flags & CF_IS_SUPERSONIC) != 0;
                    if (is_demo && !(mf & MUT_TEAM_DEMOS)) is_demo = (i1 % 2) != (i2 % 2);
                    if (is_demo) {
                        c2.flags |= CF_IS_DEMOED; c2.demo_respawn_timer = A.mut.respawn_delay;
                    } else {
                        bool ground_hit = c2.flags & CF_ON_GROUND;
                        float base = ground_hit ? curve_bump_ground(speed_towards) : curve_bump_air(speed_towards);
                        V3 hit_up = ground_hit ? col2(c2.b.rot) : v3(0, 0, 1);
                        V3 imp = vel_dir * base + hit_up * curve_bump_up(speed_towards) * A.mut.bump_force_scale;
                        c2.vel_impulse_cache += imp * UU2BT;
                    }
                    c1.car_contact_other = i2 + 1;
                    c1.car_contact_cooldown = A.mut.bump_cooldown;
                    if ((i1 % 2) != (i2 % 2)) {  // Gym.cpp:30-38: only bumps on opponents count
                        ev.bump_mask |= (1u << i1);
                        if (is_demo) ev.bump_mask |= (1u << (8 + i1));
                    }
                }
            }
        }
    }
}

// ---- narrowphase over all pairs -----------------------------------------------------------------------
// The expensive pair tests (box-triangle, sphere-triangle, box-box) are "items".  collide_all() is written against a
// narrowphase provider:
//   NarrowInline  finds and runs every item on the spot (host build; overflow fallback on the device);
//   NarrowQueued  reads the results of items that collide_queue_body() found and collide_run_item() ran earlier, on
//                 other lanes of the wavefront (rlgpu_env.hip).  Item results are merged in the order the inline
//                 provider would have produced them, so both give identical contact lists.
// The candidate / item queue itself (CollideQueue) and the BVH queries live in arena_world.h: the wheel rays share it.
// One (hitbox, triangle) pair of the car-mesh manifold: GJK on the core shapes (arena_gjk.h); where the cores themselves overlap, the
// core polytopes' minimum-translation axis from the SAT routine (deepest clipped point), pushed out by the margin.
// The two rejections that come before GJK: true = this triangle yields no contact point.
RLG_HD bool hitbox_triangle_rejected(V3 bc, const M3& R, const MeshTri& t) {
    {   // btConvexTriangleCallback::processTriangle's early out (btConvexConcaveCollisionAlgorithm.cpp:103-137): the hitbox's support vertex
        // along the triangle normal, either side, is farther from the plane than the contact threshold -> no GJK for this triangle
        const V3 v0 = v3(t.v0x, t.v0y, t.v0z);
        V3 tn = cross(v3(t.v1x, t.v1y, t.v1z) - v0, v3(t.v2x, t.v2y, t.v2z) - v0);
        tn = normalized(tn);
        const V3 h = hitbox_half();
        for (int side = 0; side < 2; side++) {
            V3 dl = tmul(R, tn);
            V3 lp = v3(dl.x >= 0.f ? h.x : -h.x, dl.y >= 0.f ? h.y : -h.y, dl.z >= 0.f ? h.z : -h.z);
            V3 wp = (R * lp) + bc;
            if (dot(tn, v0) - dot(tn, wp) > CBT_CAR) return true;
            tn *= -1.f;
        }
    }
    {   // not in the reference, and not needed for its result: a triangle that stays more than the threshold outside one of the hitbox's
        // own slabs cannot come within it of the (smaller, rounded) GJK shape either -- spares the wavefront the GJK path for the
        // neighbours of the triangle actually touched (two thirds of the GJK runs in random play)
        const V3 h = hitbox_half();
        const V3 p0 = tmul(R, v3(t.v0x, t.v0y, t.v0z) - bc), p1 = tmul(R, v3(t.v1x, t.v1y, t.v1z) - bc), p2 = tmul(R, v3(t.v2x, t.v2y, t.v2z) - bc);
        const float m = CBT_CAR + 1e-4f;
        if (fminf(p0.x, fminf(p1.x, p2.x)) > h.x + m || fmaxf(p0.x, fmaxf(p1.x, p2.x)) < -(h.x + m)) return true;
        if (fminf(p0.y, fminf(p1.y, p2.y)) > h.y + m || fmaxf(p0.y, fmaxf(p1.y, p2.y)) < -(h.y + m)) return true;
        if (fminf(p0.z, fminf(p1.z, p2.z)) > h.z + m || fmaxf(p0.z, fmaxf(p1.z, p2.z)) < -(h.z + m)) return true;
    }
    return false;
}
// `prechecked`: the caller has already applied hitbox_triangle_rejected (the device's candidate phase does, one lane per candidate, so
// that only triangles that reach GJK take a slot of the item queue -- with the game's own meshes a car near a wall overlaps dozens)
RLG_HD bool hitbox_triangle(V3 bc, const M3& R, const MeshTri& t, Cand& c, bool prechecked = false) {
#ifdef RLG_GJK_STATS
    RLG_GJK_STATS(0, 0);
#endif
    if (!prechecked && hitbox_triangle_rejected(bc, R, t)) return false;
#ifdef RLG_EXPERIMENT_GJK_TWICE   // what-if build only: every GJK run done twice, same physics -> the launch grows by what the runs cost in place
    { GjkOut g0; bool d0 = false; const float thr = CBT_CAR * (1.f + 1e-7f * (float)(t.edge_flags & 1u)); if (gjk_box_triangle(bc, R, hitbox_core(), BOX_MARGIN, t, thr, g0, d0) && g0.dist == 1234.5678f) c.dist = 0.f; }
#endif
    GjkOut g; bool deep = false;
#ifdef RLG_GJK_STATS
    RLG_GJK_STATS(1, 0);
#endif
    RLG_DBG_COUNT(8);
#ifdef RLG_ITEM_CLOCK
    const unsigned long long tg0_ = RLG_ITEM_CLOCK();
#endif
    const bool hit_ = gjk_box_triangle(bc, R, hitbox_core(), BOX_MARGIN, t, CBT_CAR, g, deep);
#ifdef RLG_ITEM_CLOCK
    RLG_SPAN_DONE(10, RLG_ITEM_CLOCK() - tg0_);
#endif
    if (hit_) {
        if (g.dist > CBT_CAR) return false;          // btManifoldResult::addContactPoint's own gate (btManifoldResult.cpp:112)
        c.n = g.n; c.pb = g.pb; c.dist = g.dist; c.n_raw = g.n; c.pa = g.pb + g.n * g.dist;
#ifdef RLG_ITEM_CLOCK
        const unsigned long long te0_ = RLG_ITEM_CLOCK();
#endif
#ifdef RLG_EXPERIMENT_ADJUST_TWICE   // what-if build only: the edge adjustment priced in place (first result unused)
        { V3 pb2 = c.pb, n2 = c.n; float d2 = c.dist * (1.f + 1e-7f * (float)(t.edge_flags & 1u)); adjust_internal_edge(t, pb2, n2, d2); if (d2 == 1234.5678f) c.dist = 0.f; }
#endif
        adjust_internal_edge(t, c.pb, c.n, c.dist);
#ifdef RLG_ITEM_CLOCK
        RLG_SPAN_DONE(12, RLG_ITEM_CLOCK() - te0_);
#endif
        return true;
    }
    if (!deep) return false;
    RLG_DBG_COUNT(9);
    Cand cs[4]; int nc = 0;
#ifdef RLG_EXPERIMENT_DEEP_TWICE   // what-if build only: the deep-penetration fallback priced in place
    { Cand c2[4]; int n2 = 0; box_triangle(bc, R, hitbox_core(), t, 1e-7f * (float)(t.edge_flags & 1u), c2, n2); if (n2 == 77) c.dist = 0.f; }
#endif
    box_triangle(bc, R, hitbox_core(), t, 0.f, cs, nc);
    if (nc == 0) return false;
    int best = 0;
    for (int q = 1; q < nc; q++) if (cs[q].dist < cs[best].dist) best = q;
    c = cs[best]; c.dist -= BOX_MARGIN; c.n_raw = c.n; c.pa = c.pb + c.n * c.dist;
    adjust_internal_edge(t, c.pb, c.n, c.dist);
    return true;
}

struct NarrowInline {
    template <int NC, class F>
    RLG_HD void ball_mesh(const Arena<NC>& A, MeshView mesh, F&& emit) {
        const float r = K::BALL_RADIUS * UU2BT;
        V3 bp = A.ball.b.pos, lo, hi;
        ball_query_aabb(bp, lo, hi);
        mesh_query(mesh, lo, hi, [&](int ti) {
            Cand c;
            if (sphere_triangle(bp, r, CBT_BALL, mesh.tris[ti], c.pb, c.n, c.dist) && !(c.dist > CBT_BALL)) { c.n_raw = c.n; c.pa = c.pb + c.n * c.dist; adjust_internal_edge(mesh.tris[ti], c.pb, c.n, c.dist); emit(c, (int)mesh.tris[ti].obj); }
        });
    }
    template <int NC, class F>
    RLG_HD void car_mesh(const Arena<NC>& A, MeshView mesh, int ci, F&& emit) {
        const Car& car = A.cars[ci];
        V3 bc, lo, hi;
        car_query_aabb(car, bc, lo, hi);
        mesh_query(mesh, lo, hi, [&](int ti) {
            Cand c;
            if (hitbox_triangle(bc, car.b.rot, mesh.tris[ti], c)) emit(c, (int)mesh.tris[ti].obj);
        });
    }
    template <int NC>
    RLG_HD void car_car(const Arena<NC>& A, int ia, int ib, Cand (&cs)[4], int& nc) {
        const Car& ca = A.cars[ia]; const Car& cb = A.cars[ib];
        // box A = the manifold's body0 = the HIGHER car (arena_contact.h); normals point from the lower car towards it, pb lies on the lower car
        box_box_ode((cb.b.rot * hitbox_off()) + cb.b.pos, cb.b.rot, (ca.b.rot * hitbox_off()) + ca.b.pos, ca.b.rot, hitbox_half(), cs, nc);
    }
};

template <int NCQ>
struct NarrowQueued {
    const CollideQueue<NCQ>& Q;
    RLG_HD int count() const { return Q.n_items < ITEM_CAP ? Q.n_items : ITEM_CAP; }
    template <int NC, class F>
    RLG_HD void ball_mesh(const Arena<NC>&, MeshView mesh, F&& emit) {
        for (int k = 0; k < count(); k++) {
            const CollideItem& it = Q.items[k];
            if (it.type != 0) continue;
            for (int q = 0; q < it.n; q++) emit(Q.pool[it.off + q], (int)mesh.tris[it.ref].obj);
        }
    }
    template <int NC, class F>
    RLG_HD void car_mesh(const Arena<NC>&, MeshView mesh, int ci, F&& emit) {
        for (int k = 0; k < count(); k++) {
            const CollideItem& it = Q.items[k];
            if (it.type != 1 || it.a != ci) continue;
            for (int q = 0; q < it.n; q++) emit(Q.pool[it.off + q], (int)mesh.tris[it.ref].obj);
        }
    }
    template <int NC>
    RLG_HD void car_car(const Arena<NC>&, int ia, int ib, Cand (&cs)[4], int& nc) {
        for (int k = 0; k < count(); k++) {
            const CollideItem& it = Q.items[k];
            if (it.type != 2 || it.a != ia || it.ref != ib) continue;
            for (int q = 0; q < it.n; q++) cs[nc++] = Q.pool[it.off + q];
            return;
        }
    }
};

// step 2: does candidate `k` become an item?  (triangle AABB vs the body's query box; car-car pairs always do)
template <int NC>
RLG_HD bool collide_test_candidate(const Arena<NC>& A, MeshView mesh, const CollideQueue<NC>& Q, int k) {
    const uint32_t cw = queue_cand(Q, k);
    if (cw == CAND_HOLE) return false;
    CollideItem it = unpack_cand(cw);
    if (it.type == 2) return true;
    V3 lo, hi, bc;
    if (it.type == 0) ball_query_aabb(A.ball.b.pos, lo, hi);
    else {
        if (!car_collides(A.cars[it.a])) return false;
        car_query_aabb(A.cars[it.a], bc, lo, hi);   // the tight hitbox box, not the ray-extended one the list was built for
    }
    const MeshTri& t = mesh.tris[it.ref];
    if (!tri_aabb_overlap(t, lo, hi)) return false;
    if (it.type == 1) return !hitbox_triangle_rejected(bc, A.cars[it.a].b.rot, t);    // (same decisions hitbox_triangle would make: no contact either way)
    {   // the ball: SphereTriangleDetector's plane test (sphere_triangle's first exit)
        const V3 v0 = v3(t.v0x, t.v0y, t.v0z);
        V3 n = cross(v3(t.v1x, t.v1y, t.v1z) - v0, v3(t.v2x, t.v2y, t.v2z) - v0);
        const float l2 = len2(n);
        if (l2 < SIMD_EPS * SIMD_EPS) return false;
        n = vdiv_bt(n, sqrtf(l2));
        const float dplane = fabsf(dot(A.ball.b.pos - v0, n));
        return dplane < K::BALL_RADIUS * UU2BT + CBT_BALL;
    }
}

// step 3: run item `slot`
template <int NC>
RLG_HD_T4 void collide_run_item(const Arena<NC>& A, MeshView mesh, int slot, CollideQueue<NC>& Q) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS(Q);
    CollideItem it = Q.items[slot];
    // a triangle item yields at most one point, kept in registers until its pool slot is known; only the (rare) car-car item has a list
    Cand one; Cand many[4]; int n = 0;
#ifdef RLG_ITEM_CLOCK
    const unsigned long long t0_ = RLG_ITEM_CLOCK();
#endif
    if (it.type == 0) {
        const float r = K::BALL_RADIUS * UU2BT;
        if (sphere_triangle(A.ball.b.pos, r, CBT_BALL, mesh.tris[it.ref], one.pb, one.n, one.dist) && !(one.dist > CBT_BALL)) { one.n_raw = one.n; one.pa = one.pb + one.n * one.dist; adjust_internal_edge(mesh.tris[it.ref], one.pb, one.n, one.dist); n = 1; }
    } else if (it.type == 1) {
        const Car& car = A.cars[it.a];
        V3 bc = car.b.pos + car.b.rot * hitbox_off();
        if (hitbox_triangle(bc, car.b.rot, mesh.tris[it.ref], one, true)) n = 1;
    } else {
        NarrowInline().car_car(A, it.a, it.ref, many, n);
    }
#ifdef RLG_ITEM_CLOCK
    RLG_ITEM_DONE(it.type, n, RLG_ITEM_CLOCK() - t0_);
#endif
    int off = n > 0 ? fetch_add(Q.n_pool, n) : 0;
    if (off + n > POOL_CAP) { Q.overflow = 1; n = 0; off = 0; RLG_DBG_COUNT(4); }
    if (it.type == 2) { for (int q = 0; q < n; q++) Q.pool[off + q] = many[q]; }
    else if (n == 1) Q.pool[off] = one;
    Q.items[slot].off = (int16_t)off; Q.items[slot].n = (int16_t)n;
}

// The contact list is produced in two steps so that the bodies of an env can work side by side:
//   collide_body   per body   the body's manifolds against the static world (mesh, then the four planes: the reference's pair order)
//                             and, for a car, its contact with the ball -- written into the body's own region of the contact
//                             list (arena_contact.h); counts in W.body_n[body] / W.ball_hit[car]
//   collide_merge  per env    the car-car pairs, the contact-added callbacks that touch other bodies, and the ORDER in which
//                             the solver visits the contacts (W.cidx)
// the end-of-algorithm refresh of a body's mesh manifolds (n_man windows; window j holds cnt[j] points from start[j] on): distances and world points
// rebuilt, points that fail the refresh removed (manifold_refresh_static), what is left of the windows moved down back to back; sets sid and
// rewrites start / cnt.  Returns the number of points left.
template <int MM>
RLG_HD int refresh_mesh_manifolds(Contact* out, int n_man, int (&start)[MM], int (&cnt)[MM], const Body& b, float breaking) {
    int n = 0;
    if constexpr (MM > 4) {
        for (int j = 0; j < n_man; j++) {
            const int c = manifold_refresh_static(out + start[j], cnt[j], b, v3(0, 0, 0), breaking);
            if (start[j] != n) for (int k = 0; k < c; k++) out[n + k] = out[start[j] + k];
            for (int k = 0; k < c; k++) out[n + k].sid = mesh_sid(j);
            start[j] = n; cnt[j] = c; n += c;
        }
    } else {
        RLG_UNROLL
        for (int j = 0; j < MM; j++) {     // (static window numbers: the arrays stay in registers)
            if (j >= n_man) break;
            const int c = manifold_refresh_static(out + start[j], cnt[j], b, v3(0, 0, 0), breaking);
            if (start[j] != n) for (int k = 0; k < c; k++) out[n + k] = out[start[j] + k];
            for (int k = 0; k < c; k++) out[n + k].sid = mesh_sid(j);
            start[j] = n; cnt[j] = c; n += c;
        }
    }
    return n;
}
// Returns false when the body's contacts do not fit the layout LY (a mesh object with points beyond the last manifold window, a plane point beyond
// the plane slots): nothing has been lost yet -- the caller redoes the env's world step with the big layout (world_step_finish_big).
template <int NC, class LY, class NW>
RLG_HD_BIG bool collide_body(Arena<NC>& A, MeshView mesh, Contact* list, int16_t* body_n, int8_t* ball_hit, int8_t (*body_obj)[LY::MESH_MANIFOLDS], int body, bool ball_asleep, NW nw) {
    RLG_ASSUME_LDS(A);
    if constexpr (!LY::IS_BIG) { RLG_ASSUME_LDS(*list); RLG_ASSUME_LDS(*body_n); RLG_ASSUME_LDS(*ball_hit); RLG_ASSUME_LDS(**body_obj); }   // (the small layout's TickWork is LDS-resident on the device)
    constexpr int MM = LY::MESH_MANIFOLDS;
    const float r = K::BALL_RADIUS * UU2BT;
    const V3 bp = A.ball.b.pos;
    Contact* out = &list[LY::body_region(body)];
    int n = 0;
    bool fits = true;
    // The mesh points arrive triangle by triangle in the reference's visiting order = object by object (arena_mesh.cpp).  One manifold per
    // object (Arena.cpp:1028-1054: one btBvhTriangleMeshShape + body per .cmf file): a new object opens the next window of <= 4 points.
    // (the window arrays are indexed by a run-time manifold number: as plain arrays they live in scratch memory -- with the small layout's two
    // windows a compare-select per access keeps them in registers; the big layout's thirty-two stay arrays)
    int m_start[MM], m_cntv[MM];
    auto wget = [&](const int (&a)[MM], int i) -> int {
        if constexpr (MM > 4) return a[i];
        else { int v = a[0]; RLG_UNROLL for (int q = 1; q < MM; q++) v = (i == q) ? a[q] : v; return v; }
    };
    auto wset = [&](int (&a)[MM], int i, int v) {
        if constexpr (MM > 4) a[i] = v;
        else { RLG_UNROLL for (int q = 0; q < MM; q++) a[q] = (i == q) ? v : a[q]; }
    };
    int m_obj = -1, n_man = 0;   // n_man: the window being filled
#pragma unroll
    for (int q = 0; q < MM; q++) { body_obj[body][q] = -1; m_start[q] = 0; m_cntv[q] = 0; }
    auto mesh_point = [&](const Body& b, const Cand& k, int obj, float breaking) -> bool {
        if (obj != m_obj) {
            if (n_man < MM && wget(m_cntv, n_man) > 0) { const int nxt = wget(m_start, n_man) + wget(m_cntv, n_man); n_man++; if (n_man < MM) { wset(m_start, n_man, nxt); wset(m_cntv, n_man, 0); } }   // the manifold filled so far is complete (one that kept no point is reused)
            m_obj = obj;
        }
        if (RLG_UNLIKELY(n_man >= MM)) { fits = false; return false; }   // one more mesh object with points than the layout has windows: the caller takes the big layout
        const int ws = wget(m_start, n_man); int wc = wget(m_cntv, n_man);
        const int slot = manifold_add_static(out + ws, wc, 4, b, k.n, k.pb, k.dist, breaking, k.pa);
        wset(m_cntv, n_man, wc);
        if (slot < 0) return false;
        body_obj[body][n_man] = (int8_t)obj;
        n = ws + wc;
        return true;
    };
    auto mesh_done = [&](const Body& b, float breaking) {   // the windows in use: [0, n_man] when the last one holds points, else [0, n_man)
        const int used = n_man < MM ? (wget(m_cntv, n_man) > 0 ? n_man + 1 : n_man) : MM;
        n = refresh_mesh_manifolds<MM>(out, used, m_start, m_cntv, b, breaking);
    };
    if (body == 0) {
        // A sleeping ball (ISLAND_SLEEPING, Arena.cpp:721-727) and the static world bodies (put to sleep by
        // btDiscreteDynamicsWorld::addRigidBody) are both inactive, so the dispatcher skips the pair
        // (btCollisionDispatcher::needsCollision): no ball-world contacts on the tick a car wakes the ball up.
        if (!ball_asleep) {
            // ball vs mesh: one manifold for the whole mesh body, a point per triangle in visiting order (btConvexConcaveCollisionAlgorithm.cpp:
            // 76-160 -> btSphereTriangleCollisionAlgorithm on the shared manifold), reduced to 4 by manifold_replace_index
            nw.ball_mesh(A, mesh, [&](const Cand& k, int obj) { mesh_point(A.ball.b, k, obj, CBT_BALL); });
            mesh_done(A.ball.b, CBT_BALL);
            for (int k = 0; k < n; k++) { out[k].a = 0; out[k].b = -1; out[k].special = 1; }
            // ball vs planes (btConvexPlaneCollisionAlgorithm.cpp:92-121): the sphere's support vertex towards the plane
            for (int i = 0; i < 4; i++) {
                V3 pn, po; world_plane_body(i, pn, po);
                V3 dir = -pn;                                          // planeInConvex.getBasis() * -planeNormal (ball basis = identity: m_noRot)
                // btSphereShape::localGetSupportingVertex: the point (0,0,0) pushed out by getMargin() (= the radius) along the normalised direction
                V3 vtx = dir * r;                                      // |dir| = 1 exactly for the axis-aligned planes
                V3 vip = vtx + (bp + (-po));
                // the ball in the basis a user state setter gave it (BallState::rotMat, constant under ArenaConfig::noBallRot; tests/golden/ballrot_golden.npz)
                if (RLG_UNLIKELY(!(A.ball.b.rot.r0.x == 1.f && A.ball.b.rot.r1.y == 1.f && A.ball.b.rot.r2.z == 1.f))) {
                    vtx = normalized(tmul(A.ball.b.rot, -pn)) * r; vip = (A.ball.b.rot * vtx) + (bp + (-po));
                }
                float dist = dot(pn, vip);
                if (!(dist < CBT_BALL)) continue;
                if (RLG_UNLIKELY(n >= LY::BALL_REGION)) { fits = false; continue; }
                V3 pb = (vip - pn * dist) + po;
                int cnt = 0;
                if (manifold_add_static(&out[n], cnt, 1, A.ball.b, pn, pb, dist, CBT_BALL) < 0) continue;
                if (manifold_refresh_static(&out[n], 1, A.ball.b, po, CBT_BALL) == 0) continue;
                out[n].a = 0; out[n].b = -1; out[n].sid = (int8_t)(1 + i); out[n].special = 1;
                n++;
            }
        }
    } else {
        const int ci = body - 1;
        Car& car = A.cars[ci];
        ball_hit[ci] = 0;
        if (car_collides(car)) {
            V3 h = hitbox_half();
            const Body cb = car.b;   // (a register copy: the loops below store contacts and car flags between their reads of the pose)
            V3 bc = cb.pos + cb.rot * hitbox_off();
            // pair order of the reference's broadphase (btRSBroadphase.cpp:393-469): the statics of the car's cell in creation order = mesh
            // bodies, then floor, ceiling, -x wall, +x wall (Arena.cpp:1036-1101); the contact-added callbacks fire in that order
            nw.car_mesh(A, mesh, ci, [&](const Cand& k, int obj) {
                if (mesh_point(cb, k, obj, CBT_CAR)) {
                    car.flags |= CF_WORLD_CONTACT; car.world_contact_normal = k.n_raw;   // Arena::_BtCallback_OnCarWorldCollision (Arena.cpp:420-427) sees the point BEFORE btAdjustInternalEdgeContacts (Arena.cpp:276-280)
                }
            });
            mesh_done(cb, CBT_CAR);
            for (int k = 0; k < n; k++) { out[k].a = (int8_t)(1 + ci); out[k].b = -1; out[k].special = 0; }
            // planes: ONE contact per plane and tick, the hitbox's support vertex towards the plane (btConvexPlaneCollisionAlgorithm.cpp:
            // 92-121; the perturbation passes are off: m_minimumPointsPerturbationThreshold = 0, btConvexPlaneCollisionAlgorithm.h:62-63)
            for (int i = 0; i < 4; i++) {
                V3 pn, po; world_plane_body(i, pn, po);
                // (no point of the hitbox is farther from its centre than |half extents| = 1.534: a plane farther than that plus the threshold from the
                // centre cannot give a point, whatever the pose -- three of the four planes for a car on the floor)
                if (dot(pn, bc + (-po)) - HITBOX_REACH >= CBT_CAR) continue;
                V3 dirl = tmul(cb.rot, -pn);                       // planeInConvex.getBasis() * -planeNormal
                V3 vtx = v3(dirl.x >= 0.f ? h.x : -h.x, dirl.y >= 0.f ? h.y : -h.y, dirl.z >= 0.f ? h.z : -h.z);   // btBoxShape::localGetSupportingVertex
                V3 vip = (cb.rot * vtx) + (bc + (-po));           // convexInPlaneTrans(vtx): origin = convex origin - plane origin
                float dist = dot(pn, vip);                             // plane constant 0 in the plane body's frame (Arena.cpp:1067-1101)
                if (!(dist < CBT_CAR)) continue;
                if (RLG_UNLIKELY(n >= LY::CAR_WORLD_MAX)) { fits = false; continue; }
                V3 pb = (vip - pn * dist) + po;                        // planeObjWrap->getWorldTransform() * vtxInPlaneProjected
                int cnt = 0;
                if (manifold_add_static(&out[n], cnt, 1, cb, pn, pb, dist, CBT_CAR) < 0) continue;
                car.flags |= CF_WORLD_CONTACT; car.world_contact_normal = pn;    // (the contact-added callback: before the algorithm's refresh)
                if (manifold_refresh_static(&out[n], 1, cb, po, CBT_CAR) == 0) continue;
                out[n].a = (int8_t)(1 + ci); out[n].b = -1; out[n].sid = (int8_t)(1 + i); out[n].special = 0;
                n++;
            }
            // car vs ball: the manifold's body0 is the car (arena_contact.h).  Its callback runs in collide_merge.
            // The pair goes through the same btGjkPairDetector as a hitbox against a triangle (btConvexConvexAlgorithm; the sphere-box
            // algorithm is not registered, btDefaultCollisionConfiguration.cpp): A = the box, B = the ball's btSphereShape, a point whose margin
            // is the radius.  Where the point gets inside the box core (never seen in play) the analytic form answers instead of EPA.
            V3 pb, pn; float dist; bool have = false;
            V3 sl, sh_, hl, hh;
            // (the hitbox's box reaches at most HITBOX_REACH from its centre along every axis, the ball's r + 0.08: farther apart than both along some axis, the boxes cannot touch)
            const V3 sep = bp - bc; const float far = HITBOX_REACH + (r + 0.08f);
            const bool maybe = !(fabsf(sep.x) > far || fabsf(sep.y) > far || fabsf(sep.z) > far);
            if (RLG_UNLIKELY(maybe)) { sphere_shape_aabb(bp, sl, sh_); hitbox_shape_aabb(cb.pos, cb.rot, hl, hh); }
            if (RLG_UNLIKELY(maybe && aabb_touch(hl, hh, sl, sh_))) {   // the child shapes' boxes must touch before the pair's algorithm runs at all (btCompoundCollisionAlgorithm.cpp:333-358)
                GjkOut g; bool deep = false;
                if (gjk_box_sphere(bc, cb.rot, hitbox_core(), BOX_MARGIN, bp, r, CBT_BALL, g, deep)) {
                    if (!(g.dist > CBT_BALL)) { pn = g.n; pb = g.pb; dist = g.dist; have = true; }
                } else if (deep && sphere_box(bp, r, bc, cb.rot, h, CBT_BALL, pb, pn, dist)) {
                    pn = -pn; pb = bp + pn * r; have = true;   // normal on the ball, pointing at the car; the point on the ball
                }
            }
            if (RLG_UNLIKELY(have)) {
                Contact& c = list[LY::car_ball_slot(ci)];
                manifold_point_dynamic(c, cb, A.ball.b, pn, pb, dist);
                c.a = (int8_t)(1 + ci); c.b = 0; c.sid = 0; c.special = 0;
                ball_hit[ci] = 1;
            }
        }
    }
    body_n[body] = (int16_t)n;
    return fits;
}

// ---- sequential-impulse solve (btSequentialImpulseConstraintSolver.cpp:795-983,1003-1211,1601-1926) ----------
struct SolverBody {
    V3 v, w, dv, dw, push, turn, ext_f, ext_t;
    M3 inv_i;
    float inv_m;
    bool active;
};
template <class IDX>
struct RowT {   // 22 words with IDX = int8_t: kept small because the device kernel holds its envs' rows in LDS
    int8_t a, b;         // body indices (b = -1: static world)
    IDX fric_of;         // friction rows: index of their normal row; -1 for normal rows
    int8_t skip;         // individual ball-world rows are not iterated (m_isSpecial), only their split impulse
    V3 n1, r1xn, r2xn, ang_a, ang_b;   // contactNormal2 is -n1 whenever b >= 0
    float jac, rhs, rhs_pen, applied, applied_push, friction;
};
using Row = RowT<int8_t>;

template <class ROW, int NB>
RLG_HD void row_setup_normal(ROW& r, const Contact& c, SolverBody (&B)[NB], V3 n, V3 ra, V3 rb, float dist, float fric, float rest, bool has_b) {
    const float dt = TICK_DT;
    SolverBody& A = B[c.a];
    r.a = (int8_t)c.a; r.b = (int8_t)(has_b ? c.b : -1);
    V3 t0 = cross(ra, n);
    r.ang_a = A.inv_i * t0;
    V3 t1 = cross(rb, n);
    r.ang_b = has_b ? (B[c.b].inv_i * (-t1)) : v3(0, 0, 0);
    float d0 = A.inv_m + dot(n, cross(r.ang_a, ra));
    float d1 = has_b ? (B[c.b].inv_m + dot(n, cross(-r.ang_b, rb))) : 0.f;
    r.jac = 1.f / (d0 + d1);
    r.n1 = n; r.r1xn = t0;
    r.r2xn = has_b ? -t1 : v3(0, 0, 0);
    V3 vel1 = A.v + cross(A.w, ra);
    V3 vel2 = has_b ? (B[c.b].v + cross(B[c.b].w, rb)) : v3(0, 0, 0);
    float rel_vel = dot(n, vel1 - vel2);
    float restitution = 0.f;
    if (!(fabsf(rel_vel) < K::RESTITUTION_VEL_THRESHOLD)) restitution = rest * -rel_vel;
    if (restitution <= 0.f) restitution = 0.f;
    r.friction = fric;
    r.applied = 0.f; r.applied_push = 0.f;
    V3 efa = A.ext_f, eta = A.ext_t;
    float v1 = dot(r.n1, A.v + efa) + dot(r.r1xn, A.w + eta);
    float v2 = has_b ? (dot(-r.n1, B[c.b].v + B[c.b].ext_f) + dot(r.r2xn, B[c.b].w + B[c.b].ext_t)) : 0.f;
    float rv = v1 + v2;
    float vel_err = restitution - rv;
    float pos_err = 0.f;
    if (dist > 0.f) pos_err = 0.f; else pos_err = -dist * K::ERP2 * (1.f / dt);
    r.rhs = vel_err * r.jac;
    r.rhs_pen = pos_err * r.jac;
    r.skip = 0; r.fric_of = -1;
}

template <class ROW, int NB>
RLG_HD void row_setup_friction(ROW& r, int normal_idx, const ROW& nr, SolverBody (&B)[NB], V3 n, V3 ra, V3 rb, bool has_b) {
    SolverBody& A = B[nr.a];
    // btSolverBody::getVelocityInLocalPointNoDelta (btSolverBody.h:133-139): includes the external impulses
    V3 vel1 = (A.v + A.ext_f) + cross(A.w + A.ext_t, ra);
    V3 vel2 = has_b ? ((B[nr.b].v + B[nr.b].ext_f) + cross(B[nr.b].w + B[nr.b].ext_t, rb)) : v3(0, 0, 0);
    V3 vel = vel1 - vel2;
    float rel_vel = dot(n, vel);
    V3 lat = vel - n * rel_vel;
    float l2 = len2(lat);
    if (l2 > SIMD_EPS) lat = lat * (1.f / sqrtf(l2));
    else { V3 q; plane_space1(n, lat, q); }
    r.a = nr.a; r.b = nr.b;
    r.friction = nr.friction; r.applied = 0.f; r.applied_push = 0.f;
    r.n1 = lat; V3 f1 = cross(ra, lat); r.r1xn = f1; r.ang_a = A.inv_i * f1;
    if (has_b) { V3 f2 = cross(rb, -lat); r.r2xn = f2; r.ang_b = B[nr.b].inv_i * f2; }
    else { r.r2xn = v3(0, 0, 0); r.ang_b = v3(0, 0, 0); }
    float d0 = A.inv_m + dot(lat, cross(r.ang_a, ra));
    float d1 = has_b ? (B[nr.b].inv_m + dot(lat, cross(-r.ang_b, rb))) : 0.f;
    r.jac = 1.f / (d0 + d1);
    float v1 = dot(r.n1, A.v + A.ext_f) + dot(r.r1xn, A.w);
    float v2 = has_b ? (dot(-r.n1, B[nr.b].v + B[nr.b].ext_f) + dot(r.r2xn, B[nr.b].w)) : 0.f;
    float rv = v1 + v2;
    r.rhs = (0.f - rv) * r.jac; r.rhs_pen = 0.f;
    r.skip = 0; r.fric_of = (int8_t)normal_idx;
}

// gResolveSingleConstraintRow{LowerLimit,Generic}_scalar_reference (btSequentialImpulseConstraintSolver.cpp:46-100), cfm = 0
// One Gauss-Seidel row update.  The row's constants and both bodies' deltas are read into locals FIRST and written back LAST: the
// row and the bodies are all floats in LDS, so with the stores in between the compiler had to assume they alias and re-read the
// bodies after storing the accumulated impulse -- a second LDS round trip on the dependent chain of the solve.  (Same arithmetic.)
// the row solvers are the reference's SSE2 variants (btSequentialImpulseConstraintSolver.cpp:102-110,149-176,207-232,317-349; SOLVER_SIMD is
// on): their three-term dot product adds x to (y + z), not (x + y) to z
RLG_HD float sdot3(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
template <class ROW, int NB>
RLG_HD float row_resolve(ROW& c, SolverBody (&B)[NB], float lo, float hi, bool lower_only) {
    SolverBody& A = B[c.a];
    const int b = c.b;
    SolverBody& Bb = B[b >= 0 ? b : 0];
    const V3 n1 = c.n1, r1xn = c.r1xn, r2xn = c.r2xn, ang_a = c.ang_a, ang_b = c.ang_b;
    const float jac = c.jac, applied = c.applied;
    const V3 a_dv = A.dv, a_dw = A.dw; const float a_im = A.inv_m;
    V3 b_dv = v3(0, 0, 0), b_dw = v3(0, 0, 0); float b_im = 0.f;
    if (b >= 0) { b_dv = Bb.dv; b_dw = Bb.dw; b_im = Bb.inv_m; }
    float delta = c.rhs;
    float dv1 = sdot3(n1, a_dv) + sdot3(r1xn, a_dw);
    float dv2 = (b >= 0) ? (sdot3(-n1, b_dv) + sdot3(r2xn, b_dw)) : 0.f;
    delta -= dv1 * jac;
    delta -= dv2 * jac;
    float sum = applied + delta, now;
    if (sum < lo) { delta = lo - applied; now = lo; }
    else if (!lower_only && sum > hi) { delta = hi - applied; now = hi; }
    else now = sum;
    c.applied = now;
    A.dv = a_dv + (n1 * a_im) * delta; A.dw = a_dw + ang_a * delta;
    if (b >= 0) { Bb.dv = b_dv + ((-n1) * b_im) * delta; Bb.dw = b_dw + ang_b * delta; }
    return delta;
}
template <class ROW, int NB>
RLG_HD float row_resolve_split(ROW& c, SolverBody (&B)[NB]) {
    if (c.rhs_pen == 0.f) return 0.f;
    SolverBody& A = B[c.a];
    const int b = c.b;
    SolverBody& Bb = B[b >= 0 ? b : 0];
    const V3 n1 = c.n1, r1xn = c.r1xn, r2xn = c.r2xn, ang_a = c.ang_a, ang_b = c.ang_b;
    const float jac = c.jac, applied = c.applied_push;
    const V3 a_p = A.push, a_t = A.turn; const float a_im = A.inv_m;
    V3 b_p = v3(0, 0, 0), b_t = v3(0, 0, 0); float b_im = 0.f;
    if (b >= 0) { b_p = Bb.push; b_t = Bb.turn; b_im = Bb.inv_m; }
    float delta = c.rhs_pen;
    float dv1 = sdot3(n1, a_p) + sdot3(r1xn, a_t);
    float dv2 = (b >= 0) ? (sdot3(-n1, b_p) + sdot3(r2xn, b_t)) : 0.f;
    delta -= dv1 * jac;
    delta -= dv2 * jac;
    float sum = applied + delta, now;
    if (sum < 0.f) { delta = 0.f - applied; now = 0.f; }
    else now = sum;
    c.applied_push = now;
    A.push = a_p + (n1 * a_im) * delta; A.turn = a_t + ang_a * delta;
    if (b >= 0) { Bb.push = b_p + ((-n1) * b_im) * delta; Bb.turn = b_t + ang_b * delta; }
    return delta;
}

// Per-env scratch of one tick.  It is a parameter (not locals) so the device kernel can place it in LDS next to the
// env's state: as stack locals these arrays are dynamically indexed and would live in scratch memory.
template <int NC, int BIG = 0>
struct TickWork {
    using LY = ContactLayout<NC, BIG>;
    using idx_t = typename LY::idx_t;
    static constexpr int NB = NC + 1;
    static constexpr int MAXC = LY::MAXC;
    static constexpr int MAXM = LY::MAXM;
    static constexpr int MAXS = LY::MAXS;          // contacts the solver takes per tick (more than that: the caller takes the big layout)
    static constexpr int MAXR = LY::MAXR;          // their normal + friction rows and the ball's averaged pair
    // The cars' tick context lives from car_tick_begin to car_pre_tick_finish, the contact list from tick_world_begin (which follows) to
    // solver_finish: the device kernels (RLG_TICKWORK_OVERLAY) keep them in the same LDS bytes; the host build keeps both (the oracle's
    // debug dumps read the wheels after the tick).
#ifdef RLG_TICKWORK_OVERLAY
    union { ContactList<MAXC> L; CarTickCtx ctx[NC]; };
#else
    ContactList<MAXC> L;
    CarTickCtx ctx[NC];
#endif
    union {
        RowT<idx_t> R[MAXR];   // solver rows: built after the contact list is complete ...
        CollideQueue<NC> Q;   // ... narrowphase items: dead by then
    };
    SolverBody B[NB];      // (directly behind the union: with the queue's dead tail one stretch that nothing uses during the narrowphase, rlgpu_env.hip)
    uint64_t pad_mask[NC];           // boost pads car i touches this tick (bit p), from pads_check_car
    bool ball_asleep;
    int8_t needs_big;                // this tick's contacts do not fit the layout (collide_body / collide_merge): the env's world step is redone with the big one
    idx_t cidx[MAXC];                // slot in L of the k-th contact in solver order (collide_merge)
    idx_t nrow[MAXS], frow[MAXS];    // solver rows of the k-th contact (normal / friction), -1 = none (solver_prepare)
    int16_t body_n[8];               // world contacts in each body's region of L (collide_body)
    int8_t bp_moved[8];              // the body's proxy changed its broadphase cell this tick (bp_history_cell)
    int8_t body_obj[8][LY::MESH_MANIFOLDS];   // the mesh object of the body's k-th mesh manifold with points, -1 = none
    int8_t ball_hit[NC];             // car i touches the ball: its contact sits in car_ball_slot(i)
    int8_t man_key[MAXM], man_cnt[MAXM]; idx_t man_val[MAXM], man_first[MAXM];   // this tick's manifolds (collide_merge)
    typename LY::stack_t man_stack[MAXM];
    int8_t touch_p[LY::PAIR_POOL], touch_q[LY::PAIR_POOL], touch_cnt[LY::PAIR_POOL]; idx_t touch_first[LY::PAIR_POOL];   // the car pairs with points this tick: bodies (1 + car, lower first), their slots in L (collide_merge)
    int16_t n_normal, n_rows;
};
// (with the default caps the narrowphase queue fits inside the solver rows it shares LDS with; bigger caps -- RLG_BODY_CAND, RLG_ITEM_CAP -- grow the union)

// Which manifolds exist this tick, in the order the island manager hands them to the solver, and with them the solver order of the
// contacts (arena_contact.h explains where each piece comes from).  Only reached when at least two manifolds carry points: proxy
// boxes (with the predicted rotation), broadphase cells, union-find and the quickSort are all that is needed to ORDER them.
template <int NC, int BIG>
RLG_HD_T4 bool collide_order(Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W, int n_touching) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    using LY = ContactLayout<NC, BIG>;
    using idx_t = typename LY::idx_t;
    constexpr int MESH_MANIFOLDS = LY::MESH_MANIFOLDS, OBJ_LISTED_MAX = LY::OBJ_LISTED_MAX;
    ContactList<LY::MAXC>& L = W.L;
    constexpr int NB = NC + 1;
    // the mesh objects (one per .cmf file): each has its own proxy -- a box and the broadphase cells it is listed in (arena_mesh.cpp)
    const int n_obj = mesh.n_tris > 0 ? (mesh.bp ? (int)mesh.bp[0] : 1) : 0;
    auto obj_box = [&](int o, V3& lo, V3& hi) {
        if (!mesh.bp) { lo = v3(-1e30f, -1e30f, -1e30f); hi = v3(1e30f, 1e30f, 1e30f); return; }
        const float* mb = reinterpret_cast<const float*>(mesh.bp + 1 + (size_t)o * 6);
        const V3 t = v3(BP_THRESHOLD, BP_THRESHOLD, BP_THRESHOLD);      // a static body's proxy is grown by the contact threshold like any other (world_plane_aabb)
        lo = v3(mb[0], mb[1], mb[2]) - t; hi = v3(mb[3], mb[4], mb[5]) + t;
    };
    auto cell_objects = [&](int ci, int cj, int ck) -> uint32_t {     // the mesh objects on the static list of a broadphase cell
        if (!mesh.bp) return n_obj > 0 ? 1u : 0u;
        return mesh.bp[1 + (size_t)n_obj * 6 + bp_cell_index(ci, cj, ck)];
    };
    // proxy boxes and cells of the dynamic bodies.  A car's exact box needs the predicted rotation (sin / cos / sqrt); the bracket
    // [inner, outer] around it decides every question below the same way in all but borderline poses, and only those pay for it.
    V3 plo[NB], phi[NB]; int cx[NB], cy[NB], cz[NB]; bool live[NB];
    live[0] = true; ball_proxy_aabb(A.ball, plo[0], phi[0]);
    for (int i = 0; i < NC; i++) {
        live[1 + i] = car_collides(A.cars[i]);
        if (!live[1 + i]) continue;
        V3 il, ih, ol, oh;
        car_proxy_bracket(A.cars[i], il, ih, ol, oh);
        int a0, a1, a2, b0, b1, b2;
        bp_cell_of(il, a0, a1, a2); bp_cell_of(ol, b0, b1, b2);
        bool same = a0 == b0 && a1 == b1 && a2 == b2;
        for (uint32_t om = same ? cell_objects(a0, a1, a2) : 0u; om && same; om &= om - 1u) {
            V3 mlo, mhi; obj_box(__builtin_ctz(om), mlo, mhi); same = aabb_touch(il, ih, mlo, mhi) == aabb_touch(ol, oh, mlo, mhi);
        }
        for (int s = 0; s < 4 && same; s++) { V3 slo, shi; world_plane_aabb(s, slo, shi, BP_THRESHOLD); same = aabb_touch(il, ih, slo, shi) == aabb_touch(ol, oh, slo, shi); }
        if (same) {   // against the other dynamic bodies: their own outer / inner boxes (the ball's box is exact)
            V3 bl, bh; ball_proxy_aabb(A.ball, bl, bh);
            same = aabb_touch(il, ih, bl, bh) == aabb_touch(ol, oh, bl, bh);
            for (int j = 0; j < NC && same; j++) {
                if (j == i || !car_collides(A.cars[j])) continue;
                V3 jil, jih, jol, joh; car_proxy_bracket(A.cars[j], jil, jih, jol, joh);
                same = aabb_touch(il, ih, jil, jih) == aabb_touch(ol, oh, jol, joh);
            }
        }
        if (same) { plo[1 + i] = il; phi[1 + i] = ih; } else car_proxy_aabb(A.cars[i], plo[1 + i], phi[1 + i]);
    }
    for (int b = 0; b < NB; b++) if (live[b]) bp_cell_of(plo[b], cx[b], cy[b], cz[b]);
    // union-find over the dynamic pairs (btSimulationIslandManager::findUnions, pair-array order; btUnionFind::unite links the root of
    // the first element under the root of the second) and the manifold list in creation order
    int8_t root[NB];
    for (int b = 0; b < NB; b++) root[b] = (int8_t)b;
    auto find = [&](int x) { while (root[x] != x) { root[x] = root[root[x]]; x = root[x]; } return x; };
    int nm = 0; bool fits = true;
    int8_t (&mkey)[LY::MAXM] = W.man_key; idx_t (&mval)[LY::MAXM] = W.man_val;   // key: body0 for now, island id later; val: manifold number
    idx_t (&mfirst)[LY::MAXM] = W.man_first; int8_t (&mcnt)[LY::MAXM] = W.man_cnt;
    for (int p = 0; p < NB; p++) {
        if (!live[p]) continue;
        // statics of the cell, in creation order; a sleeping ball makes no manifold with them (needsCollision)
        if (!(p == 0 && W.ball_asleep)) {
            V3 xlo = plo[p], xhi = phi[p];     // a car's child algorithm also needs the hitbox's own box to reach the other shape's (btCompoundCollisionAlgorithm.cpp:333-358)
            if (p > 0) hitbox_shape_aabb(A.cars[p - 1].b.pos, A.cars[p - 1].b.rot, xlo, xhi);
            const int base = LY::body_region(p), nw_ = W.body_n[p];
            int k = 0, mi = 0, n_listed = 0;
            // the mesh objects listed in the body's cell, in creation (= file) order, each a manifold of its own, with or without points; the
            // (<= 2) manifolds that DO hold points are merged in at their objects' places
            uint32_t om = cell_objects(cx[p], cy[p], cz[p]);
            for (int q = 0; q < MESH_MANIFOLDS; q++) if (W.body_obj[p][q] >= 0) om |= 1u << (W.body_obj[p][q] < BP_MAX_OBJECTS ? W.body_obj[p][q] : BP_MAX_OBJECTS - 1);
            for (; om; om &= om - 1u) {
                const int o = __builtin_ctz(om);
                V3 mlo, mhi; obj_box(o, mlo, mhi);
                bool listed = aabb_touch(plo[p], phi[p], mlo, mhi);
                if (listed && p > 0) listed = aabb_touch(xlo, xhi, mlo, mhi);
                int cnt = 0; const int first = base + k;
                if (mi < MESH_MANIFOLDS && W.body_obj[p][mi] == o) {
                    const int8_t want = mesh_sid(mi);
                    while (k < nw_ && L.c[base + k].sid == want) { k++; cnt++; }
                    mi++;
                }
                if (!(listed || cnt > 0)) continue;
                if (cnt == 0 && n_listed >= OBJ_LISTED_MAX) { fits = false; continue; }     // (more empty manifolds than the list holds: their place in the sort would be lost)
                if (nm < LY::MAXM) { mkey[nm] = (int8_t)p; mfirst[nm] = (idx_t)first; mcnt[nm] = (int8_t)cnt; nm++; n_listed++; } else fits = false;
            }
            for (int s = 1; s <= 4; s++) {
                int cnt = 0, first = base + k;
                if (k < nw_ && L.c[base + k].sid == s) { cnt = 1; k++; }
                V3 slo, shi, glo, ghi; world_plane_aabb(s - 1, slo, shi); world_plane_aabb(s - 1, glo, ghi, BP_THRESHOLD);     // the shape's box; the proxy's
                if (!aabb_touch(plo[p], phi[p], glo, ghi) || (p > 0 && !aabb_touch(xlo, xhi, slo, shi))) continue;
                if (nm < LY::MAXM) { mkey[nm] = (int8_t)p; mfirst[nm] = (idx_t)first; mcnt[nm] = (int8_t)cnt; nm++; } else fits = false;
            }
        }
        // dynamic partners filed in the same cell neighbourhood with overlapping proxy boxes (pairs are made by the lower body)
        // ... in the order of the cell's dynamic list = the order in which the proxies last ARRIVED in their cells (bp_history_track)
        int8_t qorder[NB]; int nq = 0;
        for (int q = p + 1; q < NB; q++) {
            int at = nq++;
            while (at > 0 && (A.bp_hist[qorder[at - 1]] & 7u) > (A.bp_hist[q] & 7u)) { qorder[at] = qorder[at - 1]; at--; }
            qorder[at] = (int8_t)q;
        }
        for (int qi = 0; qi < nq; qi++) {
            const int q = qorder[qi];
            if (!live[q]) continue;
            const int dx = cx[p] - cx[q], dy = cy[p] - cy[q], dz = cz[p] - cz[q];
            if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) continue;
            if (!aabb_touch(plo[p], phi[p], plo[q], phi[q])) continue;
            { const int i = find(p), j = find(q); if (i != j) root[i] = (int8_t)j; }
            // the pair's manifold exists when the children's own boxes touch (btCompoundCollisionAlgorithm.cpp:127-141,333-358)
            V3 l1, h1, l2, h2;
            if (p == 0) {
                const Car& c = A.cars[q - 1];
                sphere_shape_aabb(A.ball.b.pos, l1, h1); hitbox_shape_aabb(c.b.pos, c.b.rot, l2, h2);
                if (!aabb_touch(l2, h2, l1, h1)) continue;
                if (nm < LY::MAXM) { mkey[nm] = (int8_t)q; mfirst[nm] = (idx_t)LY::car_ball_slot(q - 1); mcnt[nm] = W.ball_hit[q - 1]; nm++; } else fits = false;
            } else {
                const Car& ca = A.cars[p - 1]; const Car& cb = A.cars[q - 1];
                hitbox_shape_aabb(ca.b.pos, ca.b.rot, l1, h1); hitbox_shape_aabb(cb.b.pos, cb.b.rot, l2, h2);
                if (!aabb_touch(l1, h1, l2, h2)) continue;
                int first = LY::PAIR_BASE, cnt = 0;
                for (int k = 0; k < n_touching; k++) if (W.touch_p[k] == p && W.touch_q[k] == q) { first = W.touch_first[k]; cnt = W.touch_cnt[k]; }
                if (nm < LY::MAXM) { mkey[nm] = (int8_t)q; mfirst[nm] = (idx_t)first; mcnt[nm] = (int8_t)cnt; nm++; } else fits = false;
            }
        }
    }
    if (!fits) return false;   // (the caller takes the big layout, whose list holds every manifold there can be)
    // island id of a manifold = island of its body0 (always dynamic here); quickSort by it
    for (int m = 0; m < nm; m++) { mkey[m] = (int8_t)find(mkey[m]); mval[m] = (idx_t)m; }
    bt_quicksort(mkey, mval, nm, W.man_stack);
    int n = 0;
    for (int m = 0; m < nm; m++) {
        const int mi = mval[m];
        for (int k = 0; k < mcnt[mi]; k++) W.cidx[n++] = (idx_t)(mfirst[mi] + k);
    }
    L.n = n;
    return true;
}

// btRSBroadphase keeps, per cell, the list of dynamic proxies filed under it and the 26 cells around it (btRSBroadphase.h:65-72).  When
// setAabb finds a proxy in a new cell it is erased from its old 27 lists and pushed back onto the new 27 (btRSBroadphase.cpp:185-203,
// 287-325), so every list is ordered by when its proxies last changed cell -- creation order to begin with, object order within a tick --
// and calculateOverlappingPairs makes a body's pairs in THAT order (:393-469).  The pair array orders the manifold array, and Bullet's
// unstable island sort turns another array into another solver order once an island holds several manifolds (a heap of cars).  This is
// the per-tick bookkeeping: the cell of every active body's proxy, and a new arrival rank for those that changed it.
// In two parts: per body (a lane each on the device, with the body's contacts) the cell, per env the ranks.
template <int NC, int BIG>
RLG_HD_SMALL void bp_history_cell(Arena<NC>& A, TickWork<NC, BIG>& W, int b) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    W.bp_moved[b] = 0;
#ifdef RLG_EXPERIMENT_NO_BP_TRACK   /* what-if build only (tools/build_variant.sh): prices the bookkeeping */
    return;
#endif
    int i, j, k;
    uint32_t h = A.bp_hist[b];
    if (h == 0u) {   // never filed (no body can be in cell 0 with rank 0): a fresh arena's proxies, created in object order (ball, cars) at
                     // their creation poses (btCollisionWorld::addCollisionObject: the shape's box, no threshold)
        V3 lo, hi;
        if (b == 0) sphere_shape_aabb(v3(0.f, 0.f, K::BALL_REST_Z * UU2BT), lo, hi);
        else compound_shape_aabb(v3(0.f, 0.f, 17.f * UU2BT), m3_identity(), lo, hi);
        bp_cell_of(lo, i, j, k);
        h = ((uint32_t)bp_cell_index(i, j, k) << 3) | (uint32_t)b;
        A.bp_hist[b] = (uint16_t)h;
    }
    // btCollisionWorld::updateAabbs visits EVERY object (m_forceUpdateAllAabbs is Bullet's default and the reference leaves it, btCollisionWorld.cpp:62,178-193): a
    // ball a state setter put down at rest somewhere else is filed under its new cell by the next tick, asleep or not (round 6: a live Gym rollout whose solver
    // order depended on it, tools/live_gym_hip.py).  A wreck keeps the proxy it had (it neither moves nor collides until Car::Respawn re-files it).
    if (b != 0 && !car_collides(A.cars[b - 1])) return;
    if (b == 0) { V3 lo, hi; ball_proxy_aabb(A.ball, lo, hi); bp_cell_of(lo, i, j, k); }
    else {       // the bracket around the proxy box names the cell without the predicted rotation in all but borderline poses
        V3 il, ih, ol, oh; car_proxy_bracket(A.cars[b - 1], il, ih, ol, oh);
        int i2, j2, k2; bp_cell_of(il, i, j, k); bp_cell_of(ol, i2, j2, k2);
        if (i != i2 || j != j2 || k != k2) { V3 lo, hi; car_proxy_aabb(A.cars[b - 1], lo, hi); bp_cell_of(lo, i, j, k); }
    }
    const uint32_t c = (uint32_t)bp_cell_index(i, j, k);
    if (c != (h >> 3)) { A.bp_hist[b] = (uint16_t)((c << 3) | (h & 7u)); W.bp_moved[b] = 1; }
}
template <int NC, int BIG>
RLG_HD_SMALL void bp_history_ranks(Arena<NC>& A, const TickWork<NC, BIG>& W) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    constexpr int NB = NC + 1;
    uint32_t moved = 0u;
    for (int b = 0; b < NB; b++) moved |= W.bp_moved[b] ? 1u << b : 0u;
    if (!moved) return;
    // those that stayed keep their order and close ranks; the movers follow in object order
    int8_t nr[NB]; int n_stay = 0;
    for (int b = 0; b < NB; b++) {
        if (moved >> b & 1u) continue;
        int r = 0;
        for (int o = 0; o < NB; o++) r += (!(moved >> o & 1u) && (A.bp_hist[o] & 7u) < (A.bp_hist[b] & 7u)) ? 1 : 0;
        nr[b] = (int8_t)r; n_stay++;
    }
    for (int b = 0; b < NB; b++) if (moved >> b & 1u) nr[b] = (int8_t)n_stay++;
    for (int b = 0; b < NB; b++) A.bp_hist[b] = (uint16_t)((A.bp_hist[b] & ~7u) | (uint32_t)nr[b]);
}

// three or more cars on the ball in one tick: the callbacks in the order the ball's pairs were made (collide_merge)
template <int NC, int BIG>
RLG_HD_COLD void ball_callbacks_by_rank(Arena<NC>& A, TickWork<NC, BIG>& W) {
    using LY = ContactLayout<NC, BIG>;
    for (int r = 0; r < NC + 1; r++)
        for (int ci = 0; ci < NC; ci++)
            if (W.ball_hit[ci] && (int)(A.bp_hist[1 + ci] & 7u) == r) on_car_ball_contact(A, ci, W.L.c[LY::car_ball_slot(ci)].rb);
}

// per env: the contact-added callbacks that touch other bodies, the car-car pairs, and the solver order of all contacts.
// Returns false -- BEFORE any callback has fired or any history has been written -- when the tick's contacts do not fit the layout (a car-car
// point beyond the pair pool, more contacts than the solver has rows for): the caller redoes the env's world step with the big layout.
template <int NC, int BIG, class NW>
RLG_HD_BIG bool collide_merge(Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W, TickEvents& ev, bool& ball_car_touch, NW nw) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    using LY = ContactLayout<NC, BIG>;
    using idx_t = typename LY::idx_t;
    ContactList<LY::MAXC>& L = W.L;
    ball_car_touch = false;
    // car-car pairs first, points only: body0 of the manifold = the higher car (arena_contact.h); the gate is conservative, the exact box test
    // follows.  (Nothing here depends on what the callbacks below change: car_collides reads the tick-start `frozen`.)
    int n_touching = 0, n_pair = 0;
    int pair_man[NC];   // manifolds with points each car shares with other cars
    for (int ci = 0; ci < NC; ci++) pair_man[ci] = 0;
    bool fits = true;
    for (int ia = 0; ia < NC && fits; ia++) {
        for (int ib = ia + 1; ib < NC && fits; ib++) {
            const Car& ca = A.cars[ia]; const Car& cb = A.cars[ib];
            if (RLG_LIKELY(!car_collides(ca) || !car_collides(cb) || !cars_maybe_touch(A, ia, ib))) continue;
            V3 l1, h1, l2, h2;
            hitbox_shape_aabb(ca.b.pos, ca.b.rot, l1, h1); hitbox_shape_aabb(cb.b.pos, cb.b.rot, l2, h2);
            if (!aabb_touch(l1, h1, l2, h2)) continue;
            Cand cs[4]; int nc = 0;
            nw.car_car(A, ia, ib, cs, nc);
            const int first = LY::PAIR_BASE + n_pair;
            int cnt = 0;
            for (int k = 0; k < nc; k++) {
                if (cs[k].dist > CBT_CAR) continue;
                if (n_pair >= LY::PAIR_POOL) { fits = false; break; }     // a car-car point beyond the pool: the big layout has room for every pair's four
                Contact& c = L.c[LY::PAIR_BASE + n_pair];
                manifold_point_dynamic(c, cb.b, ca.b, cs[k].n, cs[k].pb, cs[k].dist);
                c.a = (int8_t)(1 + ib); c.b = (int8_t)(1 + ia); c.sid = 0; c.special = 0;
                n_pair++; cnt++;
            }
            if (cnt > 0 && n_touching < LY::PAIR_POOL) {
                W.touch_p[n_touching] = (int8_t)(1 + ia); W.touch_q[n_touching] = (int8_t)(1 + ib); W.touch_first[n_touching] = (idx_t)first; W.touch_cnt[n_touching] = (int8_t)cnt; n_touching++;
                pair_man[ia]++; pair_man[ib]++;
            }
        }
    }
    int total = W.body_n[0] + n_pair;
    int n_ball = 0;
    for (int ci = 0; ci < NC; ci++) { const int bh = W.ball_hit[ci] ? 1 : 0; n_ball += bh; total += bh + W.body_n[1 + ci]; }
    if (RLG_UNLIKELY(!fits || total > LY::MAXS)) return false;
    bp_history_ranks(A, W);    // (idempotent for a given set of movers: the big layout's second run files the same ranks)
    // The order of the solver's rows only matters between rows that share a body.  Count, per dynamic body, the manifolds with points
    // that touch it: while no body has two, every row stands alone -- any order gives bit-identical results -- and slot order is used.
    int n = 0, ball_man = 0, max_man = 0;
    for (int k = 0; k < W.body_n[0]; k++) { W.cidx[n++] = (idx_t)k; ball_man += (k == 0 || L.c[k].sid != L.c[k - 1].sid); }
    int car_man[NC];
    for (int ci = 0; ci < NC; ci++) {
        const int base = LY::body_region(1 + ci);
        int cm = 0;
        if (RLG_UNLIKELY(W.ball_hit[ci])) { W.cidx[n++] = (idx_t)LY::car_ball_slot(ci); cm++; ball_man++; }
        for (int k = 0; k < W.body_n[1 + ci]; k++) { W.cidx[n++] = (idx_t)(base + k); cm += (k == 0 || L.c[base + k].sid != L.c[base + k - 1].sid); }
        car_man[ci] = cm + pair_man[ci];
    }
    if (RLG_UNLIKELY(n_touching > 0))
        for (int t = 0; t < n_touching; t++)
            for (int k = 0, first = W.touch_first[t], cnt = W.touch_cnt[t]; k < cnt; k++) W.cidx[n++] = (idx_t)(first + k);
    L.n = n;
    max_man = ball_man;
    for (int ci = 0; ci < NC; ci++) max_man = car_man[ci] > max_man ? car_man[ci] : max_man;
    // some body is held by two manifolds: the reference's pair / island order decides which of them the solver visits first.  (Nothing the
    // callbacks below change is read by it: poses, the tick-start `frozen`, the broadphase history.)
    if (RLG_UNLIKELY(max_man >= 2)) { if (!collide_order<NC, BIG>(A, mesh, W, n_touching)) return false; }
#ifdef RLG_EXPERIMENT_ORDER_TWICE   // what-if build only: prices the order emulation in place (it is idempotent)
    if (max_man >= 2) collide_order<NC, BIG>(A, mesh, W, n_touching);
#endif
    // ---- from here on the tick is committed to this layout: the contact-added callbacks that touch other bodies ----
    // ball-touch callbacks in pair order (Arena::_BtCallback_OnCarBallCollision).  Each adds its extra hit velocity to the ball's impulse cache, so from three
    // touching cars on the ORDER shows in the sum's last bit (two commute): the ball's pairs are made in the order of its cell's dynamic list, i.e. by the cars'
    // arrival ranks (bp_history_ranks above; btRSBroadphase.cpp:393-469) -- not by slot.  (Round 6, tools/random_tapes.py ... aerial, seeds 70681 / 80921: three
    // and four cars on the ball in one tick.)
    if (RLG_UNLIKELY(n_ball >= 3)) { ball_car_touch = true; ball_callbacks_by_rank<NC, BIG>(A, W); }
    else
        for (int ci = 0; ci < NC; ci++)
            if (RLG_UNLIKELY(W.ball_hit[ci])) { ball_car_touch = true; on_car_ball_contact(A, ci, L.c[LY::car_ball_slot(ci)].rb); }
    // ... and the car-car callbacks, pair by pair, point by point (Arena::_BtCallback_OnCarCarCollision(car1 = the manifold's body0 = the higher car,
    // car2): equal user indices, no swap, Arena.cpp:231-240)
    if (RLG_UNLIKELY(n_touching > 0)) for (int t = 0; t < n_touching; t++) {
        const int ia = W.touch_p[t] - 1, ib = W.touch_q[t] - 1;
        for (int k = 0, first = W.touch_first[t], cnt = W.touch_cnt[t]; k < cnt; k++) {
            const Contact c = L.c[first + k];
            on_car_car_contact(A, ib, ia, tmul(A.cars[ib].b.rot, c.ra), tmul(A.cars[ia].b.rot, c.rb), ev);
        }
    }
    return true;
}

// world step, first part (per env): sleep flag, gravity, damping; leaves an empty narrowphase queue
template <int NC, int BIG>
RLG_HD void world_step_begin(Arena<NC>& A, TickWork<NC, BIG>& W) {
    const float dt = TICK_DT;
    // ball sleep flag (Arena.cpp:721-727): taken at tick start by tick_build_candidates (nothing touches the ball before here)
    const bool ball_asleep = W.ball_asleep;
    W.needs_big = 0;
    // applyGravity (btDiscreteDynamicsWorld.cpp:265-276): active bodies only
    const V3 g = v3(A.mut.gravity_x, A.mut.gravity_y, A.mut.gravity_z) * UU2BT;   // MutatorConfig::gravity (Arena.cpp:25: btDynamicsWorld::setGravity)
    // btRigidBody::setGravity keeps acceleration * (1 / m_inverseMass) (btRigidBody.cpp:132-139): not quite mass * g in float
    if (!ball_asleep) A.ball.b.force += g * (1.0f / BALL_INV_MASS);
    for (int i = 0; i < NC; i++) if (!A.cars[i].frozen) A.cars[i].b.force += g * (1.0f / CAR_INV_MASS);
    // predictUnconstraintMotion: damping (btRigidBody.cpp:153-165); car damping is 0 -> pow(1,dt) = 1
    A.ball.b.vel *= A.mut.ball_damp_per_tick;   // btPow(1 - linearDamping, timeStep), linearDamping = MutatorConfig::ballDrag (Arena.cpp:46)
}

// the narrowphase run inline (an env whose queue overflowed this tick, 1.8 per million env-ticks; the host build without a queue): cold calls, so
// that the common path does not carry a second copy of collide_body / collide_merge
template <int NC, int BIG>
RLG_HD_COLD bool collide_body_inline(Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W, int body) {
    return collide_body<NC, typename TickWork<NC, BIG>::LY>(A, mesh, W.L.c, W.body_n, W.ball_hit, W.body_obj, body, W.ball_asleep, NarrowInline());
}
template <int NC, int BIG>
RLG_HD_COLD bool collide_merge_inline(Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W, TickEvents& ev, bool& touch) {
    return collide_merge<NC, BIG>(A, mesh, W, ev, touch, NarrowInline());
}
// world step, second part, in four pieces of different width (the host runs them back to back):
//   solver_body_contacts per body  the body's region of the contact list (`queued`: from the narrowphase queue W.Q, else / on overflow inline)
//   solver_prepare   per env       contact list merged in reference order + callbacks + car-car pairs, solver
//                                  bodies, the averaged ball-world contact, and which solver rows each contact gets
//   solver_rows      per contact   its normal row and its friction row (btSequentialImpulseConstraintSolver.cpp:1003-1211)
//   solver_iterate   per env       split-impulse + velocity iterations (:1601-1877) -- sequential by nature (Gauss-Seidel)
//   solver_finish    per body      write back (:1878-1904), integrateTransforms (btDiscreteDynamicsWorld.cpp:889-1027), clearForces
template <int NC, int BIG>
RLG_HD_MID void solver_prepare(Arena<NC>& A, MeshView mesh, TickEvents& ev, TickWork<NC, BIG>& W, bool queued) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    const float dt = TICK_DT;
    constexpr int NB = NC + 1;
    constexpr int MAXC = TickWork<NC, BIG>::MAXC;
    ContactList<MAXC>& L = W.L;
    SolverBody (&B)[NB] = W.B;
    auto& R = W.R;
    const bool ball_asleep = W.ball_asleep;
#ifdef RLG_QSTAT
    RLG_QSTAT(W.Q);   // host-side statistics hook of tools (queue fill levels before the rows overwrite them)
#endif

    bool touch = false, fits;
    if (RLG_UNLIKELY(W.needs_big)) return;      // (a body's contacts did not fit: collide_body)
    if (RLG_LIKELY(queued && !W.Q.overflow)) fits = collide_merge<NC, BIG>(A, mesh, W, ev, touch, NarrowQueued<NC>{W.Q});
    else fits = collide_merge_inline(A, mesh, W, ev, touch);
    if (RLG_UNLIKELY(!fits)) { W.needs_big = 1; return; }   // (with the big layout: cannot happen, every pair and every slot has a row)
    const bool ball_active = !ball_asleep || touch;  // island woken by an active car (btSimulationIslandManager.cpp)
#ifdef RLG_PROF_SPLIT_PREPARE
    RLG_PROF(7);
#endif
    RLG_SPROF(44);

    // (the solver bodies were filled by solver_body_setup, a lane per body; what the merge decides is whether the ball's island is awake)
    B[0].active = ball_active;
    RLG_SPROF(45);
    // row numbering: normal rows in solver order, then the averaged special row, then one friction row per non-special
    // contact, then the special row's friction row
    int nr = 0, n_special = 0; V3 sp_normal = v3(0, 0, 0); float sp_dist = 0.f;
    for (int k = 0; k < L.n; k++) {
        const Contact& c = L.c[W.cidx[k]];
        W.nrow[k] = (int8_t)nr++; W.frow[k] = -1;
        if (c.special) { n_special++; sp_normal += c.n; sp_dist += len(c.ra); }
    }
    const int n_contact_rows = nr;
    if (n_special > 0) nr++;
    W.n_normal = nr;
    for (int k = 0; k < L.n; k++) if (!L.c[W.cidx[k]].special) W.frow[k] = (int8_t)nr++;
    if (n_special > 0) {  // convertContactSpecial (btSequentialImpulseConstraintSolver.cpp:1164-1211): its two rows are set up here
        Contact spc;
        float distance = sp_dist / (float)n_special;
        V3 normal = vdiv_bt(sp_normal, (float)n_special);
        spc.a = 0; spc.b = -1; spc.sid = 0; spc.special = 0; spc.n = normal; spc.dist = distance; spc.ra = normal * -distance; spc.rb = v3(0, 0, 0);
        row_setup_normal(R[n_contact_rows], spc, B, spc.n, spc.ra, spc.rb, spc.dist, ball_world_friction(A.mut), ball_world_restitution(A.mut), false);
        row_setup_friction(R[nr], n_contact_rows, R[n_contact_rows], B, spc.n, spc.ra, spc.rb, false);
        nr++;
    }
    W.n_rows = nr;
}

// the solver's view of one body (btSequentialImpulseConstraintSolver.cpp:initSolverBody): velocities, inverse mass / inertia, the external
// impulses of this tick's forces.  Per body (a lane each on the device), between the body's contacts and solver_prepare; `active` of the ball
// is settled by solver_prepare (a car that touches it wakes its island).
template <int NC, int BIG>
RLG_HD_SMALL void solver_body_setup(const Arena<NC>& A, TickWork<NC, BIG>& W, int body) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    const float dt = TICK_DT;
    SolverBody& s = W.B[body];
    const Body& b = body == 0 ? A.ball.b : A.cars[body - 1].b;
    const float inv_m = body == 0 ? BALL_INV_MASS : CAR_INV_MASS;
    s.v = b.vel; s.w = b.angvel; s.dv = s.dw = s.push = s.turn = v3(0, 0, 0);
    s.inv_m = inv_m; s.inv_i = b.inv_inertia_w;
    s.ext_f = b.force * inv_m * dt; s.ext_t = tmul(b.inv_inertia_w, b.torque) * dt;
    s.active = body == 0 ? true : !A.cars[body - 1].frozen;
}

// contacts of one body (see collide_body); `queued` as in solver_prepare
template <int NC, int BIG>
RLG_HD_SMALL void solver_body_contacts(Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W, int body, bool queued) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    using LY = typename TickWork<NC, BIG>::LY;
    bp_history_cell(A, W, body);
    RLG_SPROF(43);
    bool fits;
    if (RLG_LIKELY(queued && !W.Q.overflow)) fits = collide_body<NC, LY>(A, mesh, W.L.c, W.body_n, W.ball_hit, W.body_obj, body, W.ball_asleep, NarrowQueued<NC>{W.Q});
    else fits = collide_body_inline(A, mesh, W, body);
    if (RLG_UNLIKELY(!fits)) W.needs_big = 1;   // (cleared by tick_world_begin; the bodies of an env may all store the same 1)
}

template <int NC, int BIG>
RLG_HD void solver_rows(const Mutators& mut, TickWork<NC, BIG>& W, int k) {
    const Contact& c = W.L.c[W.cidx[k]];
    const int ni = W.nrow[k];
    if (ni < 0) return;
    row_setup_normal(W.R[ni], c, W.B, c.n, c.ra, c.rb, c.dist, contact_friction(c, mut), contact_restitution(c, mut), c.b >= 0);
    if (c.special) W.R[ni].skip = 1;
    const int fi = W.frow[k];
    if (fi >= 0) row_setup_friction(W.R[fi], ni, W.R[ni], W.B, c.n, c.ra, c.rb, c.b >= 0);
}

// (Tried and dropped: the velocity deltas of all bodies in registers for the whole solve, a body index becoming a compare chain over
// the NB slots -- it removes the two LDS round trips per row from the dependent chain, but the chains are ~100 more VALU instructions
// per row and with one wavefront per SIMD the kernel is bound by instructions issued, not by LDS latency: 553 K -> 882 K cycles.)
// (Tried and dropped: one lane per body when no row joins two bodies -- such rows commute exactly -- with a group vote for the
// split-impulse early exit.  The envs that are slow here have ball-car / car-car rows and stay one sequence, and the per-row
// chain lookup cost more than the short chains saved: 483 K -> 570 K cycles per launch on the slowest workgroup.)
template <int NC, int BIG>
RLG_HD_SMALL void solver_iterate(TickWork<NC, BIG>& W) {
    RLG_ASSUME_LDS_W(W);
    constexpr int NB = NC + 1;
    SolverBody (&B)[NB] = W.B;
    auto& R = W.R;
    const int n_normal = W.n_normal, nr = W.n_rows;
    RLG_PROF(3);
    // split-impulse iterations
    for (int it = 0; it < K::SOLVER_ITERS; it++) {
        float resid = 0.f;
        for (int k = 0; k < n_normal; k++) {
            float d = row_resolve_split(R[k], B);
            float rr = d * (1.f / R[k].jac);
            resid = fmaxf(resid, rr * rr);
        }
        if (resid <= 0.f || it >= K::SOLVER_ITERS - 1) break;
    }
    // velocity iterations.  (Tried and dropped: fetching row k+1 while row k computes -- the extra register copies cost more issue
    // slots than the fetch latency they hid: 467 K -> 537 K cycles on the slowest workgroup.)
    // An iteration in which every row's impulse change is exactly zero leaves the bodies and the accumulated impulses as they were
    // (x + 0 = x; the deltas start at +0 and never hold a -0), so every later iteration would repeat it: stopping there gives the
    // reference's ten iterations' result bit for bit.  A ball rolling on the floor -- one normal and one friction row -- gets there
    // in three or four.
    for (int it = 0; it < K::SOLVER_ITERS; it++) {
        bool moved = false;
        for (int k = 0; k < n_normal; k++) if (!R[k].skip) moved |= row_resolve(R[k], B, 0.f, 1e10f, true) != 0.f;
        for (int k = n_normal; k < nr; k++) {
            float total = R[R[k].fric_of].applied;
            if (total > 0.f) {
                moved |= row_resolve(R[k], B, -(R[k].friction * total), R[k].friction * total, false) != 0.f;
            }
        }
        if (!moved) break;
    }
    RLG_PROF(4);
}

template <int NC, int BIG>
RLG_HD_SMALL void solver_finish(Arena<NC>& A, TickWork<NC, BIG>& W, int body) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    if constexpr (BIG == 0) { if (RLG_UNLIKELY(W.needs_big)) return; }   // (the env's world step was finished by world_step_finish_big)
    const float dt = TICK_DT;
    const SolverBody s = W.B[body];   // (register copies of the solver body and of the rigid body: see car_wheel_ray_finish)
    if (body == 0) {
        Body b = A.ball.b;
        if (s.active) {
            b.vel = (s.v + s.dv) + s.ext_f; b.angvel = (s.w + s.dw) + s.ext_t;
            if (!is_zero(s.push) || !is_zero(s.turn)) b.pos = b.pos + s.push * dt;  // m_noRot: orientation untouched
            b.pos = b.pos + b.vel * dt;
            A.ball.b.vel = b.vel; A.ball.b.angvel = b.angvel; A.ball.b.pos = b.pos;
        }
        A.ball.b.force = v3(0, 0, 0); A.ball.b.torque = v3(0, 0, 0);
    } else {
        Car& c = A.cars[body - 1];
        if (s.active) {
            const uint32_t flags = c.flags;
            Body b = c.b;
            b.vel = (s.v + s.dv) + s.ext_f; b.angvel = (s.w + s.dw) + s.ext_t;
            // A car demolished by this tick's contact callback: the reference's reported rotation is the copy Car::_PostTickUpdate takes
            // (Car.cpp:135-138), which it skips for a demoed car -- so the state keeps the pre-tick basis while position and velocities
            // come from the body (Car.cpp:10-20).  The body is disabled from the next pre-tick on and rebuilt at respawn.
            // The BODY turns all the same, and stays in the world as it then stands: car_ghost_rot (arena_world.h).
            const bool rot_stale = RLG_UNLIKELY((flags & CF_IS_DEMOED) != 0);
            M3 rot = b.rot;
            if (RLG_UNLIKELY(!is_zero(s.push) || !is_zero(s.turn))) {
                b.pos = b.pos + s.push * dt;
                rot = integrate_rotation(rot, s.turn * K::SPLIT_TURN_ERP, dt);
            }
            b.pos = b.pos + b.vel * dt;
            RLG_SPROF(46);
            rot = integrate_rotation(rot, b.angvel, dt);
            RLG_SPROF(48);
            if (rot_stale) b.inv_inertia_w = rot;
            else {
                b.rot = rot;
                body_update_inertia(b, car_inv_inertia_local());
            }
            b.force = v3(0, 0, 0); b.torque = v3(0, 0, 0);
            c.b = b;
        } else {
            c.b.force = v3(0, 0, 0); c.b.torque = v3(0, 0, 0);
        }
    }
}

// The world step of an env whose contacts did not fit the small layout, redone from the contacts on with the big one (arena_contact.h: a manifold
// for every mesh object, a slot for every car pair's four points, a solver row for every slot).  Nothing of the first attempt has left a trace
// that the second does not write again: collide_body's only stores into the arena are the car's world-contact flag and normal (same points, same
// order, now all of them), collide_merge gave up before its first callback and before the broadphase ranks.  The broadphase CELLS were filed by
// the first attempt (bp_history_cell is not idempotent): `moved` hands them over.  One lane on the device, Wb in global memory.
template <int NC>
RLG_HD_NOINLINE void world_step_finish_big(Arena<NC>& A, MeshView mesh, TickEvents& ev, bool ball_asleep, const int8_t* moved, TickWork<NC, 1>& Wb) {
    using LY = ContactLayout<NC, 1>;
    Wb.ball_asleep = ball_asleep; Wb.needs_big = 0;
    for (int b = 0; b <= NC; b++) Wb.bp_moved[b] = moved[b];
    bool fits = true;
    for (int body = 0; body <= NC; body++) fits &= collide_body<NC, LY>(A, mesh, Wb.L.c, Wb.body_n, Wb.ball_hit, Wb.body_obj, body, ball_asleep, NarrowInline());
    for (int body = 0; body <= NC; body++) solver_body_setup(A, Wb, body);
    solver_prepare(A, mesh, ev, Wb, false);
    if (!fits || Wb.needs_big) RLG_DBG_COUNT(7);   // (not reachable: the big layout has room for whatever the mesh format can describe)
    for (int k = 0; k < Wb.L.n; k++) solver_rows(A.mut, Wb, k);
    solver_iterate(Wb);
    for (int body = 0; body <= NC; body++) solver_finish(A, Wb, body);
}

// (host form of the world step's second part; `big`: where the fallback may put its work -- the host allocates it when it is needed)
template <int NC, int BIG>
RLG_HD void world_step_finish(Arena<NC>& A, MeshView mesh, TickEvents& ev, TickWork<NC, BIG>& W, bool queued, TickWork<NC, 1>* big = nullptr) {
    for (int body = 0; body <= NC; body++) solver_body_contacts(A, mesh, W, body, queued);
    for (int body = 0; body <= NC; body++) solver_body_setup(A, W, body);
    solver_prepare(A, mesh, ev, W, queued);
    if constexpr (BIG == 0) {
        if (W.needs_big) {
            RLG_DBG_COUNT(10);
#if !defined(__HIP_DEVICE_COMPILE__)
            TickWork<NC, 1>* own = big ? nullptr : new TickWork<NC, 1>;
            world_step_finish_big(A, mesh, ev, W.ball_asleep, W.bp_moved, big ? *big : *own);
            delete own;
#else
            world_step_finish_big(A, mesh, ev, W.ball_asleep, W.bp_moved, *big);
#endif
            RLG_PROF(5);
            return;
        }
    }
    for (int k = 0; k < W.L.n; k++) solver_rows(A.mut, W, k);
    solver_iterate(W);
    for (int body = 0; body <= NC; body++) solver_finish(A, W, body);
    RLG_PROF(5);
}

// ---- boost pads (BoostPad.cpp:51-105, BoostPadGrid.cpp:5-25; locations RLConst.h:215-253) -----------------
RLG_HD V3 pad_pos(int i) {
    const float BIG[6][2] = {{-3584.f, 0.f}, {3584.f, 0.f}, {-3072.f, 4096.f}, {3072.f, 4096.f}, {-3072.f, -4096.f}, {3072.f, -4096.f}};
    const float SM[28][2] = {{0.f, -4240.f}, {-1792.f, -4184.f}, {1792.f, -4184.f}, {-940.f, -3308.f}, {940.f, -3308.f}, {0.f, -2816.f},
        {-3584.f, -2484.f}, {3584.f, -2484.f}, {-1788.f, -2300.f}, {1788.f, -2300.f}, {-2048.f, -1036.f}, {0.f, -1024.f}, {2048.f, -1036.f},
        {-1024.f, 0.f}, {1024.f, 0.f}, {-2048.f, 1036.f}, {0.f, 1024.f}, {2048.f, 1036.f}, {-1788.f, 2300.f}, {1788.f, 2300.f},
        {-3584.f, 2484.f}, {3584.f, 2484.f}, {0.f, 2816.f}, {-940.f, 3308.f}, {940.f, 3308.f}, {-1792.f, 4184.f}, {1792.f, 4184.f}, {0.f, 4240.f}};
    if (i < 6) return v3(BIG[i][0], BIG[i][1], 73.f);
    return v3(SM[i - 6][0], SM[i - 6][1], 70.f);
}

// BoostPadGrid::pads[8][10]: one pad per cell (BoostPadGrid.cpp:27-41); cell = (int)(pos / 1024 + half)
RLG_HD int pad_of_cell(int cell) {
    const int8_t CELL_PAD[80] = {-1, -1, 12, -1, -1, 0, -1, 26, -1, -1, -1, 4, -1, -1, -1, -1, -1, -1, -1, 2, 7, -1, 14, 16, -1, -1, 21, 24, -1, 31,
        -1, 9, -1, -1, -1, 19, -1, -1, 29, -1, 6, 10, 11, -1, 17, -1, 22, 28, 30, 33, 8, -1, 15, -1, -1, 20, -1, 25, -1, 32,
        -1, -1, -1, 18, -1, -1, 23, -1, -1, -1, -1, 5, 13, -1, -1, 1, -1, 27, -1, 3};
    return CELL_PAD[cell];
}
// the boost pad lookup words (PAD_TAB_WORDS; filled once on the host, staged in LDS by the kernels -- a table behind a computed index
// in constant memory costs a ~400-cycle global load per lookup; handed to the tick as a pointer of its own: one more member in MeshView
// pushes that by-value argument onto the stack at every call).  A car whose position falls into cell
// (ix, iy) is tested against the pads of the 3 x 3 cells around it (BoostPadGrid.cpp:5-25); of those only a pad within 300 uu of the
// cell's own extent can pass either test (cylinder radius <= 208 uu about the car's origin; locked-pad box half width <= 160 uu against
// the car's box, which reaches <= 89 uu from its origin), so the others are left out: at most three remain.
inline void pad_table_fill(uint32_t* out) {
    for (int p = 0; p < 34; p++) { V3 pp = pad_pos(p); out[p] = (uint32_t)((int)pp.x + 8192) | ((uint32_t)((int)pp.y + 8192) << 16); }
    const float REACH = 300.f;
    for (int ix = 0; ix < 8; ix++) for (int iy = 0; iy < 10; iy++) {
        // positions that truncate to this cell: (int)(x / 1024 + 4) == ix  (cell 0 also takes the values in (-1, 0))
        const float x0 = (float)((ix == 0 ? -1 : ix) - 4) * 1024.f, x1 = (float)(ix + 1 - 4) * 1024.f;
        const float y0 = (float)((iy == 0 ? -1 : iy) - 5) * 1024.f, y1 = (float)(iy + 1 - 5) * 1024.f;
        uint32_t w = 0; int n = 0;
        for (int cx = ix - 1; cx <= ix + 1; cx++) for (int cy = iy - 1; cy <= iy + 1; cy++) {   // same cell order as the reference's loops
            if (cx < 0 || cx > 7 || cy < 0 || cy > 9) continue;
            const int p = pad_of_cell(cx * 10 + cy);
            if (p < 0) continue;
            const V3 pp = pad_pos(p);
            if (pp.x < x0 - REACH || pp.x > x1 + REACH || pp.y < y0 - REACH || pp.y > y1 + REACH) continue;
            if (n < 3) w |= (uint32_t)(p + 1) << (8 * n);
            n++;
        }
        out[34 + ix * 10 + iy] = n <= 3 ? w : 0xFFFFFFFFu;   // (never more than three with the soccar layout; 0xFFFFFFFF = use the plain loops)
    }
}

RLG_HD const uint32_t* pad_table_default() {
#if defined(__HIP_DEVICE_COMPILE__)
    return nullptr;
#else
    static const struct PadTab { uint32_t w[PAD_TAB_WORDS]; PadTab() { pad_table_fill(w); } } t;   // (thread-safe one-time fill)
    return t.w;
#endif
}

// one pad against one car: BoostPad::CheckCollide (BoostPad.cpp:51-81) -- the cylinder about the car's origin, or for the car that holds
// the pad's lock the pad's box against the car's box
RLG_HD bool pad_touches_car(int p, V3 pad_uu, bool locked_by_car, V3 car_pos, V3 cmin, V3 cmax) {
    const bool big = p < 6;
    V3 pbt = pad_uu * UU2BT;
    if (locked_by_car) {
        float br = (big ? K::PAD_BOX_RAD_BIG : K::PAD_BOX_RAD_SMALL) * UU2BT;
        V3 bmin = pbt - v3(br, br, 0), bmax = pbt + v3(br, br, K::PAD_BOX_HEIGHT * UU2BT);
        return (bmax.x > cmin.x && bmax.y > cmin.y && bmax.z > cmin.z) && (bmin.x < cmax.x && bmin.y < cmax.y && bmin.z < cmax.z);
    }
    float rad = (big ? K::PAD_CYL_RAD_BIG : K::PAD_CYL_RAD_SMALL) * UU2BT;
    float dx = car_pos.x - pbt.x, dy = car_pos.y - pbt.y;
    if (dx * dx + dy * dy < rad * rad) return fabsf(car_pos.z - pbt.z) < (K::PAD_CYL_HEIGHT * UU2BT);
    return false;
}

// which pads does car `ci` touch (bit p)?  Reads the pads' prev_locked only, so cars can be checked in any order; the
// caller then sets cur_locked in car order (a later car overrides an earlier one, as the reference's loop does).
// (The cell tables used to sit in constant memory behind computed indices: up to 9 + 4 dependent ~400-cycle loads, 8 % of a tick.)
// car box for the locked-pad test (btCompoundShape::getAabb: centre + |R| * half extents)
RLG_HD void pad_car_box(const Car& car, V3& cmin, V3& cmax) {
    V3 h = hitbox_half();
    V3 bc = car.b.pos + car.b.rot * hitbox_off();
    V3 ext = v3(h.x * fabsf(car.b.rot.r0.x) + h.y * fabsf(car.b.rot.r0.y) + h.z * fabsf(car.b.rot.r0.z),
                h.x * fabsf(car.b.rot.r1.x) + h.y * fabsf(car.b.rot.r1.y) + h.z * fabsf(car.b.rot.r1.z),
                h.x * fabsf(car.b.rot.r2.x) + h.y * fabsf(car.b.rot.r2.y) + h.z * fabsf(car.b.rot.r2.z));
    cmin = bc - ext; cmax = bc + ext;
}
// the reference's loops as they stand (BoostPadGrid.cpp:5-25): the 3 x 3 cells around the car's
template <int NC>
RLG_HD uint64_t pads_check_car_cells(const Arena<NC>& A, int ci, int ix, int iy) {
    const Car& car = A.cars[ci];
    V3 cmin, cmax; pad_car_box(car, cmin, cmax);
    uint64_t mask = 0;
    int lox = ix - 1 < 0 ? 0 : ix - 1, hix = ix + 1 > 7 ? 7 : ix + 1, loy = iy - 1 < 0 ? 0 : iy - 1, hiy = iy + 1 > 9 ? 9 : iy + 1;
    for (int cx = lox; cx <= hix; cx++) for (int cy = loy; cy <= hiy; cy++) {
        int p = pad_of_cell(cx * 10 + cy);
        if (p < 0) continue;
        if (pad_touches_car(p, pad_pos(p), A.pads[p].prev_locked == ci + 1, car.b.pos, cmin, cmax)) mask |= (1ull << p);
    }
    return mask;
}
template <int NC>
RLG_HD_SMALL uint64_t pads_check_car(const Arena<NC>& A, const uint32_t* tab, int ci) {
    RLG_ASSUME_LDS(A);
    const Car& car = A.cars[ci];
    uint64_t mask = 0;
    if (RLG_UNLIKELY((car.flags & CF_IS_DEMOED) || car.boost >= 100)) return mask;
    V3 cp = car.b.pos * BT2UU;
    if (cp.z > K::PAD_CYL_HEIGHT + 250.f) return mask;
    int ix = (int)(cp.x / 1024 + 4), iy = (int)(cp.y / 1024 + 5);
    // no table (single-lane device callers of arena_tick: none on the product path) or a car outside the pad grid (never in play)
    if (RLG_UNLIKELY(!tab || ix < 0 || ix > 7 || iy < 0 || iy > 9)) return pads_check_car_cells(A, ci, ix, iy);
#if defined(__HIP_DEVICE_COMPILE__)
    RLG_ASSUME_LDS(*tab);
#endif
    const uint32_t near_pads = tab[34 + ix * 10 + iy];
    if (near_pads == 0u) return mask;
    if (RLG_UNLIKELY(near_pads == 0xFFFFFFFFu)) return pads_check_car_cells(A, ci, ix, iy);   // a pad layout with a crowded cell
    V3 cmin, cmax; pad_car_box(car, cmin, cmax);
    RLG_NOUNROLL
    for (uint32_t w = near_pads; w != 0u; w >>= 8) {
        const int p = (int)(w & 0xffu) - 1;
        const uint32_t word = tab[p];
        const V3 pp = v3((float)((int)(word & 0xffffu) - 8192), (float)((int)(word >> 16) - 8192), p < 6 ? 73.f : 70.f);
        if (pad_touches_car(p, pp, A.pads[p].prev_locked == ci + 1, car.b.pos, cmin, cmax)) mask |= (1ull << p);
    }
    return mask;
}

// per-pad halves of the tick (BoostPad.cpp:37-105)
RLG_HD void pad_pre_tick(Pad& pd) {
    if (pd.cooldown > 0) pd.cooldown = fmaxf(pd.cooldown - TICK_DT, 0.f);
    pd.is_active = (pd.cooldown == 0.f);
    pd.cur_locked = 0;
}
RLG_HD bool pad_gives_boost(const Pad& pd) { return pd.cur_locked != 0 && pd.is_active; }
template <int NC>
RLG_HD void pad_post_tick(Arena<NC>& A, int p) {
    Pad& pd = A.pads[p];
    int locked = 0;
    if (pd.cur_locked) {
        locked = pd.cur_locked;
        if (pd.is_active) {
            Car& c = A.cars[locked - 1];
            c.boost = fminf(c.boost + (p < 6 ? K::PAD_BOOST_BIG : K::PAD_BOOST_SMALL), K::BOOST_MAX);
            pd.is_active = false;
            pd.cooldown = p < 6 ? A.mut.pad_cooldown_big : A.mut.pad_cooldown_small;   // BoostPad.cpp:100
        }
    }
    pd.prev_locked = (int8_t)locked;
}
template <int NC>
RLG_HD void pads_lock(Arena<NC>& A, int ci, uint64_t mask) {
    for (int p = 0; p < 34; p++) if ((mask >> p) & 1ull) A.pads[p].cur_locked = (int8_t)(ci + 1);
}

// ---- Arena::Step, one tick (Arena.cpp:716-812) ---------------------------------------------------------
// The tick is a fixed sequence of phases; each phase is a set of independent work items (cars, wheels, or the env).
// arena_tick() below runs them in loops (host build, single-lane device callers); rlgpu_env.hip runs the same phase
// functions with one wavefront lane per work item.

// phase 3, per env: boost pad cooldowns (unless the caller spread them over lanes), then the first part of the dynamics world step
template <int NC, int BIG>
RLG_HD_SMALL void tick_world_begin(Arena<NC>& A, TickWork<NC, BIG>& W, bool pads_done) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS_W(W);
    if (!pads_done) for (int p = 0; p < 34; p++) pad_pre_tick(A.pads[p]);
    world_step_begin(A, W);
    RLG_PROF(0);
}
// phases 3b (lane per candidate: AABB test + compaction into items) and 3c (lane per item), host form
template <int NC>
RLG_HD void collide_compact_and_run(const Arena<NC>& A, MeshView mesh, CollideQueue<NC>& Q) {
    if (Q.overflow) return;
    int n = 0;
    auto consider = [&](int k) {
        if (!collide_test_candidate(A, mesh, Q, k)) return;
        if (n < ITEM_CAP) Q.items[n] = unpack_cand(queue_cand(Q, k)); else Q.overflow = 1;
        n++;
    };
    for (int body = 0; body <= NC; body++) for (int i = 0; i < Q.cand_count[body]; i++) consider(CollideQueue<NC>::region(body) + i);
    for (int i = 0; i < Q.n_pairs; i++) consider(CollideQueue<NC>::PAIR_BASE + i);
    if (Q.overflow) return;
    Q.n_items = n;
    for (int slot = 0; slot < n; slot++) collide_run_item(A, mesh, slot, Q);
}
// phase 3d, per env: rest of the world step = world_step_finish

// phase 4, per car: Car::_PostTickUpdate + _FinishPhysicsTick
template <int NC>
RLG_HD_SMALL void tick_car_post(Arena<NC>& A, int i) {
    RLG_ASSUME_LDS(A);
    // touch only the fields the post tick needs through locals (see car_pre_tick_finish about aliasing)
    Car& cr = A.cars[i];
    Car c; c.flags = cr.flags; c.b.vel = cr.b.vel; c.b.angvel = cr.b.angvel; c.supersonic_time = cr.supersonic_time;
    c.car_contact_cooldown = cr.car_contact_cooldown; c.ctl = cr.ctl; c.last = cr.last; c.vel_impulse_cache = cr.vel_impulse_cache;
    car_post_tick(c);
    cr.flags = c.flags; cr.b.vel = c.b.vel; cr.b.angvel = c.b.angvel; cr.supersonic_time = c.supersonic_time;
    cr.car_contact_cooldown = c.car_contact_cooldown; cr.last = c.last; cr.vel_impulse_cache = c.vel_impulse_cache;
}

// phase 5, per env: boost pad pickups (in car order), ball finish, tick counter.  `pads_done`: the caller ran
// pads_check_car / pads_lock / pad_post_tick over lanes already.
template <int NC>
RLG_HD_SMALL void tick_finish(Arena<NC>& A, const uint32_t* pad_tab, bool pads_done) {
    RLG_ASSUME_LDS(A);
    if (!pads_done) {
        RLG_NOUNROLL
        for (int k = 0; k < NC; k++) { const int i = car_at_rank(A, k); pads_lock(A, i, pads_check_car(A, pad_tab, i)); }
        for (int p = 0; p < 34; p++) pad_post_tick(A, p);
    }
    {   // Ball::_FinishPhysicsTick (Ball.cpp:112-138)
        Ball& b = A.ball;
        if (RLG_UNLIKELY(!is_zero(b.vel_impulse_cache))) { b.b.vel += b.vel_impulse_cache; b.vel_impulse_cache = v3(0, 0, 0); }
        const float vmax = A.mut.ball_max_speed * UU2BT;   // Ball.cpp:126
        if (RLG_UNLIKELY(len2(b.b.vel) > vmax * vmax)) b.b.vel = normalized(b.b.vel) * vmax;
        if (RLG_UNLIKELY(len2(b.b.angvel) > K::BALL_MAX_ANG_SPEED * K::BALL_MAX_ANG_SPEED)) b.b.angvel = normalized(b.b.angvel) * K::BALL_MAX_ANG_SPEED;
        A.ball_update_counter++;
    }
    A.tick_count++;
    RLG_PROF(6);
}

// phase 0b, per env (host form; the device walks the BVH with a lane per frontier node): sleep flag + this tick's candidates
template <int NC, int BIG>
RLG_HD void tick_build_candidates(const Arena<NC>& A, MeshView mesh, TickWork<NC, BIG>& W) {
    W.ball_asleep = (len2(A.ball.b.vel) == 0.f && len2(A.ball.b.angvel) == 0.f);
    collide_build_candidates(A, mesh, W.ball_asleep, W.Q);
}

template <int NC, int BIG>
RLG_HD void arena_tick(Arena<NC>& A, MeshView mesh, uint32_t seed, uint32_t env_id, TickEvents& ev, TickWork<NC, BIG>& W) {
    bool ref_due = false;
    for (int i = 0; i < NC; i++) ref_due = car_tick_begin(A, i, seed, env_id) || ref_due;
    if (ref_due) cars_respawn_ref_engine(A);
    tick_build_candidates(A, mesh, W);
    for (int i = 0; i < NC; i++) for (int w = 0; w < 4; w++) car_wheel_ray_begin(A, i, w, W.ctx[i]);
    if (!W.Q.overflow)
        for (int i = 0; i < NC; i++) for (int pr = 0, n = car_ray_pairs(A, W.Q, i); pr < n; pr++) car_ray_pair(A, mesh, W.Q, i, pr, W.ctx[i]);
    for (int i = 0; i < NC; i++) for (int w = 0; w < 4; w++) car_wheel_ray_finish(A, i, w, mesh, W.Q, W.ctx[i]);
    for (int k = 0; k < NC; k++) { const int i = car_at_rank(A, k); car_pre_tick_finish(A, i, W.ctx[i]); }   // Arena.cpp:716-812: the reference's car order
    tick_world_begin(A, W, false);
    collide_compact_and_run(A, mesh, W.Q);
    world_step_finish(A, mesh, ev, W, true);
    for (int i = 0; i < NC; i++) tick_car_post(A, i);
    tick_finish(A, pad_table_default(), false);
}

}  // namespace rlg
