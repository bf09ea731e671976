// arena_contact.h — one tick's contact manifolds: how contact points are cached per pair of bodies, which pairs exist, and the
// order in which the solver visits them.
//
// The reference (RocketSim's patched Bullet 3.24) keeps NO contact state from one tick to the next: its broadphase
// (BulletCollision/BroadphaseCollision/btRSBroadphase.cpp:393-469, the default: ArenaConfig.h:36 useCustomBroadphase = true)
// removes every overlapping pair at the start of calculateOverlappingPairs and adds the current ones again, and removing a pair
// destroys its collision algorithm and releases the algorithm's btPersistentManifold (btOverlappingPairCache.cpp:229-247 ->
// cleanOverlappingPair).  Every manifold the solver sees was therefore filled during this tick's narrowphase, every point has
// m_appliedImpulse = 0 (so SOLVER_USE_WARMSTARTING adds nothing, btSequentialImpulseConstraintSolver.cpp:935-942) and
// m_lifeTime = 1 (checked on the compiled reference with oracle/ref_driver.cpp:ref_debug_manifolds).  What DOES carry
// structure is, per tick:
//   * one manifold per overlapping pair, filled in the order its algorithm reports points, at most 4 points, a fifth one
//     replacing the cached point that btPersistentManifold::sortCachedPoints picks (getCacheEntry is patched to "always add",
//     btPersistentManifold.cpp:191-195);
//   * the order of the manifolds in the solver = creation order of this tick's pairs, then btSimulationIslandManager's
//     quickSort by island id (btSimulationIslandManager.cpp:137-143,386) -- not a stable sort, and manifolds WITHOUT points
//     take part in it;
//   * the end-of-algorithm refreshContactPoints that rebuilds world positions and the distance from the local points.
// This file restates those three things for the fixed body set {ball, NC cars} against {mesh, floor, ceiling, -x wall, +x wall}.
#pragma once
#include <type_traits>
#include "arena_body.h"

namespace rlg {

struct Contact {      // 44 B (LDS-resident on the device: 24..56 of them per env)
    V3 ra, rb;        // contact point on body a / b relative to that body's origin, world axes (rb: world point for a static b)
    V3 n;             // m_normalWorldOnB: points from b towards a
    float dist;
    int8_t a, b;      // the manifold's body0 / body1: 0 = ball, 1 + i = car i, -1 = static world (b only).  Which of the two
                      // is body0 follows the reference's algorithm nesting: {ball, car} vs static -> a = the dynamic body;
                      // car vs ball -> a = car, b = ball (btCompoundCollisionAlgorithm swapped + btConvexConvexAlgorithm(box, sphere));
                      // car i vs car j, i < j -> a = car j, b = car i (two nested compound algorithms, the inner one swapped)
    int8_t sid;       // static body when b == -1: 0 = the (first) triangle mesh object the body touches, 1..4 = floor, ceiling, -x wall, +x wall
                      // (creation order, Arena.cpp:1036-1101), 5.. = further mesh objects (mesh_sid; one manifold per .cmf file: Arena.cpp:1028-1054)
    int8_t special;   // ball-world contact: resolved through one averaged row (Arena.cpp:265-273)
};
// m_combinedFriction / m_combinedRestitution after the contact-added callback (Arena.cpp:283-427): car-ball and car-car points get RLConst's values, car-world points
// MutatorConfig's (Arena.cpp:425-426); a ball-world point keeps what btManifoldResult combined from the two bodies (btManifoldResult.cpp:56-78: against a static
// body the smaller friction and the larger restitution; the arena's bodies have 0.6 / 0.3, Arena.cpp:505-506; the ball's are MutatorConfig's, Arena.cpp:44-45)
RLG_HD float ball_world_friction(const Mutators& m) { const float a = m.ball_world_friction, b = K::WORLD_FRICTION; return (a < b) ? a : b; }
RLG_HD float ball_world_restitution(const Mutators& m) { const float a = m.ball_world_restitution, b = K::WORLD_RESTITUTION; return (a > b) ? a : b; }
RLG_HD float contact_friction(const Contact& c, const Mutators& m) {
    return c.b < 0 ? (c.a == 0 ? ball_world_friction(m) : m.car_world_friction) : (c.b == 0 ? K::CARBALL_FRICTION : K::CARCAR_FRICTION);
}
RLG_HD float contact_restitution(const Contact& c, const Mutators& m) {
    return c.b < 0 ? (c.a == 0 ? ball_world_restitution(m) : m.car_world_restitution) : (c.b == 0 ? K::CARBALL_RESTITUTION : K::CARCAR_RESTITUTION);
}

template <int MAXC>
struct ContactList {
    Contact c[MAXC];
    int n;            // contacts the solver visits (their slots, in solver order: TickWork::cidx)
};

// ---- where the contacts of an env live -----------------------------------------------------------------------------------
// Fixed regions, so the bodies of an env can write side by side (collide_body, one lane per body on the device).  Two sets of capacities:
//
//   BIG = 0 (the device's LDS-resident TickWork)                        BIG = 1 (the fallback's TickWork in global memory; the host build)
//   ball          [0, 10)      <= 2 mesh manifolds (one per mesh object     [0, 132)   one manifold for EVERY mesh object there can be (BP_MAX_OBJECTS),
//                              touched, <= 4 points each), then <= 2 plane             then all four planes
//                              manifolds of one point each
//   car i         the same against the world, + 1 slot: its contact with the ball
//   car pairs     PAIR_POOL points, <= 4 per touching pair                  4 points for every pair there is
//   solver        MAXS contacts, 2 (MAXS + 1) rows                          every slot of the list
//
// The small set holds every tick of ordinary play; the few that do not fit -- a third mesh object with points on one body, a car-car point
// beyond the pool, more contacts than solver rows: two events in 393 M env-ticks of learned 3v3 (profiles/r04h_soak.txt) -- used to LOSE
// those points.  Since round 5 such a tick is detected before any contact callback has fired and the env's world step is redone from its
// contacts on with the big set (arena_step.h:world_step_finish_big): nothing is dropped, by construction.
constexpr int BP_MAX_OBJECTS = 32;   // mesh objects (.cmf files) a cell's listing mask tells apart (the game's soccar set has 16)
template <int NC, int BIG = 0> struct ContactLayout {
    static constexpr int IS_BIG = BIG;
#ifdef RLG_TINY_LAYOUT   // test builds only: a small layout that overflows in ordinary play, so that the fallback is what the suites exercise
    static constexpr int MESH_MANIFOLDS = BIG ? BP_MAX_OBJECTS : 1;
    static constexpr int OBJ_LISTED_MAX = BIG ? BP_MAX_OBJECTS : 4;
    static constexpr int PLANE_SLOTS = BIG ? 4 : 1;
    static constexpr int PAIR_POOL_SMALL = 1, MAXS_SMALL = 3;
#else
    static constexpr int PAIR_POOL_SMALL = NC == 2 ? 4 : (NC == 4 ? 8 : 12);   // (a six-car heap: nine points in one tick, `3v3_kickoff` tick 318 under another car order)
    static constexpr int MAXS_SMALL = 8 + 6 * NC;
    static constexpr int MESH_MANIFOLDS = BIG ? BP_MAX_OBJECTS : 2;    // mesh manifolds WITH points of one body
    static constexpr int OBJ_LISTED_MAX = BIG ? BP_MAX_OBJECTS : 4;    // mesh manifolds WITHOUT points that still take part in the island sort
    static constexpr int PLANE_SLOTS = BIG ? 4 : 2;
#endif
    static constexpr int CAR_WORLD_MAX = 4 * MESH_MANIFOLDS + PLANE_SLOTS, BALL_REGION = CAR_WORLD_MAX, CAR_REGION = CAR_WORLD_MAX + 1;
    static constexpr int PAIR_POOL = BIG ? (NC * (NC - 1) / 2) * 4 : PAIR_POOL_SMALL;
    static constexpr int PAIR_BASE = BALL_REGION + CAR_REGION * NC;
    static constexpr int MAXC = PAIR_BASE + PAIR_POOL;
    // manifolds of one tick at most: every dynamic body against <= OBJ_LISTED_MAX mesh objects (with or without points) + 4 planes, every
    // car against the ball, every car pair
    static constexpr int MAXM = (OBJ_LISTED_MAX + MESH_MANIFOLDS + 4) * (NC + 1) + NC + NC * (NC - 1) / 2;
    static constexpr int MAXS = BIG ? MAXC : MAXS_SMALL;   // contacts the solver takes per tick
    static constexpr int MAXR = 2 * ((BIG ? MAXS : 8 + 6 * NC) + 1);   // their normal + friction rows and the ball's averaged pair (the small layout's rows keep their size in the cut-down test build: other things borrow those LDS bytes)
    using idx_t = typename std::conditional<BIG != 0, int16_t, int8_t>::type;        // a slot / row / manifold number
    using stack_t = typename std::conditional<BIG != 0, uint32_t, uint16_t>::type;   // a (lo, hi) range of bt_quicksort
    RLG_HD static int body_region(int body) { return body == 0 ? 0 : BALL_REGION + CAR_REGION * (body - 1); }
    RLG_HD static int car_ball_slot(int ci) { return BALL_REGION + CAR_REGION * ci + CAR_WORLD_MAX; }
};
// Contact::sid of a body's mi-th mesh manifold with points (0 = the first; 1..4 are the planes)
RLG_HD int8_t mesh_sid(int mi) { return (int8_t)(mi == 0 ? 0 : 4 + mi); }

// ---- one manifold being filled (btManifoldResult::addContactPoint -> btPersistentManifold::addManifoldPoint) -------------
// While a manifold fills up, its points sit in `pts[0..4)` with ra = m_localPointA (body a's frame) and rb = the world point on b.
// btPersistentManifold::sortCachedPoints (btPersistentManifold.cpp:113-187, gContactCalcArea3Points = true): the deepest point is
// never evicted; of the others the one whose removal keeps the largest (new point, three cached points) cross-product area goes.
RLG_HD int manifold_replace_index(const Contact* pts, V3 new_local, float new_dist) {
    int max_pen = -1; float mp = new_dist;
    for (int i = 0; i < 4; i++) if (pts[i].dist < mp) { max_pen = i; mp = pts[i].dist; }
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    const V3 p0 = pts[0].ra, p1 = pts[1].ra, p2 = pts[2].ra, p3 = pts[3].ra;
    if (max_pen != 0) res[0] = len2(cross(new_local - p1, p3 - p2));
    if (max_pen != 1) res[1] = len2(cross(new_local - p0, p3 - p2));
    if (max_pen != 2) res[2] = len2(cross(new_local - p0, p3 - p1));
    if (max_pen != 3) res[3] = len2(cross(new_local - p0, p2 - p1));
    int best = -1; float mv = -1e18f;   // btVector4::closestAxis4 = absolute4().maxAxis4(): first maximum wins
    for (int i = 0; i < 4; i++) { float v = fabsf(res[i]); if (v > mv) { best = i; mv = v; } }
    return best < 0 ? 0 : best;
}
// Adds a point (normal n on b, world point pb on b, depth; pa = the world point on a as the detector reported it) of body `a` against a
// STATIC body; returns the slot it went to, -1 if rejected (btManifoldResult.cpp:112-115).  `cap` <= 4 slots are available at pts.
RLG_HD int manifold_add_static(Contact* pts, int& count, int cap, const Body& a, V3 n, V3 pb, float depth, float breaking, V3 pa);
RLG_HD int manifold_add_static(Contact* pts, int& count, int cap, const Body& a, V3 n, V3 pb, float depth, float breaking) {
    return manifold_add_static(pts, count, cap, a, n, pb, depth, breaking, pb + n * depth);
}
RLG_HD int manifold_add_static(Contact* pts, int& count, int cap, const Body& a, V3 n, V3 pb, float depth, float breaking, V3 pa) {
    if (depth > breaking) return -1;
    V3 la = tmul(a.rot, pa - a.pos);     // invXform into body a's frame
    int slot = count;
    if (count >= cap) slot = cap == 4 ? manifold_replace_index(pts, la, depth) : cap - 1;
    else count++;
    pts[slot].ra = la; pts[slot].rb = pb; pts[slot].n = n; pts[slot].dist = depth;
    return slot;
}
// What the solver reads (btPersistentManifold::refreshContactPoints, btPersistentManifold.cpp:245-256, run at the end of every
// narrowphase algorithm): world positions rebuilt from the local points, the distance from those.  Turns a filling-state point
// into its final form (ra relative to a's origin).  b_origin: origin of the static body b (pure translations, Arena.cpp:1060-1101).
RLG_HD void manifold_finish_static(Contact& c, const Body& a, V3 b_origin) {
    V3 wa = (a.rot * c.ra) + a.pos;
    V3 wb = (c.rb - b_origin) + b_origin;     // localPointB = invXform(pb), then trB(localPointB); identity basis
    c.dist = dot(wa - wb, c.n);
    c.ra = wa - a.pos; c.rb = wb;
}
// The same for a whole manifold of `count` points, INCLUDING the refresh's second pass (btPersistentManifold.cpp:272-301): a point whose refreshed
// distance exceeds the contact breaking threshold, or whose two world points have drifted apart sideways by more than it, is removed -- and
// removeContactPoint (btPersistentManifold.h:164-185) fills the hole with the LAST point, so the survivors' order changes.  Within the tick the
// transforms have not moved, so only a point the detector reported at the very edge of the threshold is affected: its distance, recomputed from
// the local points, lands one rounding on the other side.  (Seen on the tessellated arena: two nearly coplanar fillet triangles, the first
// point accepted at depth <= 0.040624548 and refreshed to 0.04062467.)  The callbacks that fired when the point was added stay fired.
// Returns the number of points left.
RLG_HD int manifold_refresh_static(Contact* pts, int count, const Body& a, V3 b_origin, float breaking) {
    // one pass, last point first: the refresh's first loop touches every point by itself, and when its second loop looks at point i the points
    // behind it are final already -- so a removed point's place is taken by the finished last one, as removeContactPoint does it
    for (int i = count - 1; i >= 0; i--) {
        const V3 n = pts[i].n;
        const V3 wa = (a.rot * pts[i].ra) + a.pos;
        const V3 wb = (pts[i].rb - b_origin) + b_origin;
        const float dist = dot(wa - wb, n);
        bool drop = !(dist <= breaking);
        if (!drop) {
            const V3 projected = wa - n * dist;
            const V3 diff = wb - projected;
            drop = dot(diff, diff) > breaking * breaking;
        }
        if (drop) {
            if (i != count - 1) pts[i] = pts[count - 1];
            count--;
        } else {
            pts[i].ra = wa - a.pos; pts[i].rb = wb; pts[i].dist = dist;
        }
    }
    return count;
}
// both bodies dynamic: pa_w / pb_w are the world points the algorithm reported
RLG_HD void manifold_point_dynamic(Contact& c, const Body& a, const Body& b, V3 n, V3 pb_w, float depth) {
    V3 pa_w = pb_w + n * depth;
    V3 la = tmul(a.rot, pa_w - a.pos), lb = tmul(b.rot, pb_w - b.pos);
    V3 wa = (a.rot * la) + a.pos, wb = (b.rot * lb) + b.pos;
    c.n = n; c.dist = dot(wa - wb, n);
    c.ra = wa - a.pos; c.rb = wb - b.pos;
}

// ---- broadphase boxes (btCollisionWorld::updateSingleAabb, btCollisionWorld.cpp:143-176) ---------------------------------
// A dynamic body's proxy box is its shape box at the current transform grown by gContactBreakingThreshold = 0.02, united (m_useContinuous)
// with the same box at the transform predictUnconstraintMotion integrated from the pre-solve velocities.
constexpr float BP_THRESHOLD = 0.02f;
RLG_HD void sphere_shape_aabb(V3 center, V3& lo, V3& hi) {   // btSphereShape::getAabb with the +0.08 patch (btSphereShape.cpp:55)
    const float ext = (K::BALL_RADIUS * UU2BT) + 0.08f;
    lo = center - v3(ext, ext, ext); hi = center + v3(ext, ext, ext);
}
RLG_HD V3 abs_rows_dot(const M3& r, V3 h) {
    return v3(h.x * fabsf(r.r0.x) + h.y * fabsf(r.r0.y) + h.z * fabsf(r.r0.z), h.x * fabsf(r.r1.x) + h.y * fabsf(r.r1.y) + h.z * fabsf(r.r1.z),
              h.x * fabsf(r.r2.x) + h.y * fabsf(r.r2.y) + h.z * fabsf(r.r2.z));
}
RLG_HD void hitbox_shape_aabb(V3 pos, const M3& rot, V3& lo, V3& hi) {   // btBoxShape::getAabb at the child's world transform (btTransformAabb)
    V3 c = (rot * hitbox_off()) + pos, e = abs_rows_dot(rot, hitbox_half());
    lo = c - e; hi = c + e;
}
RLG_HD void compound_shape_aabb(V3 pos, const M3& rot, V3& lo, V3& hi) {   // btCompoundShape::getAabb (margin 0) over the child's local box
    V3 lmin = hitbox_off() - hitbox_half(), lmax = hitbox_off() + hitbox_half();
    V3 lhe = (lmax - lmin) * 0.5f, lc = (lmax + lmin) * 0.5f;
    V3 c = (rot * lc) + pos, e = abs_rows_dot(rot, lhe);
    lo = c - e; hi = c + e;
}
RLG_HD V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
RLG_HD V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
RLG_HD bool aabb_touch(V3 lo1, V3 hi1, V3 lo2, V3 hi2) {   // TestAabbAgainstAabb2
    return !(lo1.x > hi2.x || hi1.x < lo2.x || lo1.z > hi2.z || hi1.z < lo2.z || lo1.y > hi2.y || hi1.y < lo2.y);
}
RLG_HD void ball_proxy_aabb(const Ball& b, V3& lo, V3& hi) {
    const V3 t = v3(BP_THRESHOLD, BP_THRESHOLD, BP_THRESHOLD);
    V3 l1, h1, l2, h2;
    sphere_shape_aabb(b.b.pos, l1, h1); l1 = l1 - t; h1 = h1 + t;
    sphere_shape_aabb(b.b.pos + b.b.vel * TICK_DT, l2, h2); l2 = l2 - t; h2 = h2 + t;
    lo = vmin(l1, l2); hi = vmax(h1, h2);
}
RLG_HD_T7 void car_proxy_aabb(const Car& c, V3& lo, V3& hi) {
    const V3 t = v3(BP_THRESHOLD, BP_THRESHOLD, BP_THRESHOLD);
    V3 l1, h1, l2, h2;
    compound_shape_aabb(c.b.pos, c.b.rot, l1, h1); l1 = l1 - t; h1 = h1 + t;
    M3 r2 = integrate_rotation(c.b.rot, c.b.angvel, TICK_DT);     // btRigidBody::predictIntegratedTransform
    compound_shape_aabb(c.b.pos + c.b.vel * TICK_DT, r2, l2, h2); l2 = l2 - t; h2 = h2 + t;
    lo = vmin(l1, l2); hi = vmax(h1, h2);
}
// The same box bracketed without the predicted rotation: [in_lo, in_hi] = the current-transform part (contained in the proxy box),
// [out_lo, out_hi] = it grown by how far any hitbox point can travel in one tick (contains the proxy box).  A question the two
// brackets answer alike -- which cell, does it reach a plane's half space / another proxy -- needs no sin / cos.
RLG_HD void car_proxy_bracket(const Car& c, V3& in_lo, V3& in_hi, V3& out_lo, V3& out_hi) {
    const V3 t = v3(BP_THRESHOLD, BP_THRESHOLD, BP_THRESHOLD);
    compound_shape_aabb(c.b.pos, c.b.rot, in_lo, in_hi); in_lo = in_lo - t; in_hi = in_hi + t;
    const float reach = len(c.b.vel) * TICK_DT + len(c.b.angvel) * TICK_DT * (len(hitbox_half()) + len(hitbox_off())) * 1.01f + 1e-4f;
    out_lo = in_lo - v3(reach, reach, reach); out_hi = in_hi + v3(reach, reach, reach);
}
// the voxel a proxy is filed under: cell of its box's minimum corner (btRSBroadphase.h:90-108; grid from ArenaConfig.h:23-31)
constexpr float BP_CELL = 370.f * UU2BT;
constexpr int BP_CELLS_X = 25, BP_CELLS_Y = 33, BP_CELLS_Z = 7;   // ceil((maxPos - minPos) / cell), (-4500,-6000,0)..(4500,6000,2500) uu
constexpr int BP_WORDS = (BP_CELLS_X * BP_CELLS_Y * BP_CELLS_Z + 31) / 32;
RLG_HD void bp_cell_of(V3 lo, int& i, int& j, int& k) {
    const V3 mn = v3(-4500.f * UU2BT, -6000.f * UU2BT, 0.f * UU2BT);
    const float inv = 1.f / BP_CELL;          // btVector3::operator/(scalar) multiplies by the reciprocal
    V3 f = (lo - mn) * inv;
    i = (int)f.x; j = (int)f.y; k = (int)f.z;
    i = i < 0 ? 0 : (i > BP_CELLS_X - 1 ? BP_CELLS_X - 1 : i);
    j = j < 0 ? 0 : (j > BP_CELLS_Y - 1 ? BP_CELLS_Y - 1 : j);
    k = k < 0 ? 0 : (k > BP_CELLS_Z - 1 ? BP_CELLS_Z - 1 : k);
}
RLG_HD int bp_cell_index(int i, int j, int k) { return i * BP_CELLS_Y * BP_CELLS_Z + j * BP_CELLS_Z + k; }

// ---- btAlignedObjectArray::quickSortInternal (LinearMath/btAlignedObjectArray.h), on (key, payload) pairs -----------------
// The island manager sorts the manifolds by island id with this; elements with EQUAL keys get swapped around, so the exact
// procedure matters.  Iterative: the two sub-ranges of a partition are disjoint, their order of treatment is free.
template <class VT, class ST>
RLG_HD void bt_quicksort(int8_t* key, VT* val, int n, ST* stack) {
    constexpr int SH = sizeof(ST) * 4;   // a range (lo, hi) in one stack word: half its bits each
    if (n <= 1) return;
    int sp = 0;
    stack[sp++] = (ST)(((ST)0 << SH) | (ST)(n - 1));
    while (sp > 0) {
        const ST r = stack[--sp];
        const int lo = (int)(r >> SH), hi = (int)(r & (((ST)1 << SH) - 1));
        int i = lo, j = hi;
        const int8_t x = key[(lo + hi) / 2];
        do {
            while (key[i] < x) i++;
            while (x < key[j]) j--;
            if (i <= j) {
                int8_t t = key[i]; key[i] = key[j]; key[j] = t;
                VT u = val[i]; val[i] = val[j]; val[j] = u;
                i++; j--;
            }
        } while (i <= j);
        if (lo < j) stack[sp++] = (ST)(((ST)lo << SH) | (ST)j);
        if (i < hi) stack[sp++] = (ST)(((ST)i << SH) | (ST)hi);
    }
}

}  // namespace rlg
