// arena_mesh.cpp — see arena_mesh.h
#include "arena_mesh.h"
#include "arena_contact.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <array>
#include <map>
#include <tuple>
#include <queue>

namespace rlg {

namespace {
struct P3 { float x, y, z; };
P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
P3 crs(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float dt3(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct BuildNode { float mn[3], mx[3]; int left = -1, right = -1, first = 0, count = 0; };

void tri_bounds(const MeshTri& t, float* mn, float* mx) {
    mn[0] = std::min({t.v0x, t.v1x, t.v2x}); mx[0] = std::max({t.v0x, t.v1x, t.v2x});
    mn[1] = std::min({t.v0y, t.v1y, t.v2y}); mx[1] = std::max({t.v0y, t.v1y, t.v2y});
    mn[2] = std::min({t.v0z, t.v1z, t.v2z}); mx[2] = std::max({t.v0z, t.v1z, t.v2z});
}

// ---- the reference's triangle visiting order --------------------------------------------------------------------------------------
// A convex body is collided with the mesh triangles in the order btBvhTriangleMeshShape::processAllTriangles reports them
// (btBvhTriangleMeshShape.cpp:115-117), and that order decides the order of a manifold's points and which point a fifth one replaces.
// It is a property of the tree btOptimizedBvh::build makes (btOptimizedBvh.cpp:30-170, quantized: RocketSim.cpp:167 passes
// useQuantizedAabbCompression = true), restated here:
//  * quantization frame: the mesh's vertex box (btTriangleMeshShape::recalcLocalAabb, margin 0: btConcaveShape.cpp:21) through
//    setQuantizationValues (btQuantizedBvh.cpp:73-106);
//  * one leaf per triangle: its box (zero extents widened, btOptimizedBvh.cpp:103-119) quantized to 16 bits, min rounded down to even,
//    max rounded up to odd (btQuantizedBvh.h:331-358);
//  * buildTree (btQuantizedBvh.cpp:116-188): split axis = largest variance of the DEQUANTIZED box centres (calcSplittingAxis :279-304),
//    leaves with centre > mean swapped to the front (sortAndCalcSplittingIndex :218-277), the middle instead when that leaves less than
//    a third on one side; nodes are laid out depth-first;
//  * the query (walkStacklessQuantizedTreeCacheFriendly :655-674) goes through the SUBTREE HEADERS in the order they were made
//    (updateSubtreeHeaders :190-216: when a subtree exceeds 2048 bytes = 128 nodes, each child that does not gets a header, and that
//    happens after both children are built) and walks each header's nodes in array order.  So the visiting order is the depth-first
//    order with the children swapped wherever the left child is small (<= 64 leaves) and the right one is not.
// All arithmetic is the reference's, operation by operation (its x86 build runs btVector3's SSE forms, btScalar.h:216-223, but everything
// used here -- differences, products with a scalar, component-wise quotients -- is element-wise there too, so the results are the same).
struct QBvhFrame {
    float mn[3], mx[3], q[3];
    void quantize(uint16_t* out, const float* p, bool is_max) const {
        for (int a = 0; a < 3; a++) {
            float v = (p[a] - mn[a]) * q[a];
            out[a] = is_max ? (uint16_t)(((uint16_t)(v + 1.f)) | 1) : (uint16_t)(((uint16_t)(v)) & 0xfffe);
        }
    }
    void unquantize(const uint16_t* in, float* out) const {
        for (int a = 0; a < 3; a++) { float v = (float)in[a] / q[a]; v += mn[a]; out[a] = v; }
    }
    void set(const float* lo, const float* hi) {   // setQuantizationValues(lo, hi, quantizationMargin = 1)
        float size[3];
        for (int a = 0; a < 3; a++) { mn[a] = lo[a] - 1.f; mx[a] = hi[a] + 1.f; size[a] = mx[a] - mn[a]; q[a] = 65533.f / size[a]; }
        uint16_t w[3]; float v[3];
        quantize(w, mn, false); unquantize(w, v);
        for (int a = 0; a < 3; a++) mn[a] = std::min(mn[a], v[a] - 1.f);
        for (int a = 0; a < 3; a++) { size[a] = mx[a] - mn[a]; q[a] = 65533.f / size[a]; }
        quantize(w, mx, true); unquantize(w, v);
        for (int a = 0; a < 3; a++) mx[a] = std::max(mx[a], v[a] + 1.f);
        for (int a = 0; a < 3; a++) { size[a] = mx[a] - mn[a]; q[a] = 65533.f / size[a]; }
    }
};
struct QLeaf { uint16_t mn[3], mx[3]; int tri; };
struct QTree {
    const QBvhFrame& F; std::vector<QLeaf>& L;
    struct Node { int left = -1, right = -1, first = 0, count = 0; };   // leaf range [first, first + count) of L
    std::vector<Node> nodes;
    void centre(int i, float* c) const {
        float lo[3], hi[3]; F.unquantize(L[i].mn, lo); F.unquantize(L[i].mx, hi);
        for (int a = 0; a < 3; a++) c[a] = 0.5f * (hi[a] + lo[a]);
    }
    void means_of(int s, int e, float* means) const {
        means[0] = means[1] = means[2] = 0.f;
        for (int i = s; i < e; i++) { float c[3]; centre(i, c); for (int a = 0; a < 3; a++) means[a] += c[a]; }
        const float inv = 1.f / (float)(e - s);
        for (int a = 0; a < 3; a++) means[a] *= inv;
    }
    int split_axis(int s, int e) const {
        float means[3], var[3] = {0.f, 0.f, 0.f};
        means_of(s, e, means);
        for (int i = s; i < e; i++) { float c[3]; centre(i, c); for (int a = 0; a < 3; a++) { float d = c[a] - means[a]; d = d * d; var[a] += d; } }
        const float inv = 1.f / ((float)(e - s) - 1.f);
        for (int a = 0; a < 3; a++) var[a] *= inv;
        return var[0] < var[1] ? (var[1] < var[2] ? 2 : 1) : (var[0] < var[2] ? 2 : 0);   // btVector3::maxAxis
    }
    int split_index(int s, int e, int axis) {
        float means[3]; means_of(s, e, means);
        const float split = means[axis];
        int k = s;
        for (int i = s; i < e; i++) { float c[3]; centre(i, c); if (c[axis] > split) { std::swap(L[i], L[k]); k++; } }
        const int n = e - s, third = n / 3;
        if (k <= s + third || k >= e - 1 - third) k = s + (n >> 1);
        return k;
    }
    int build(int s, int e) {
        const int idx = (int)nodes.size();
        nodes.push_back(Node{});
        nodes[idx].first = s; nodes[idx].count = e - s;
        if (e - s == 1) return idx;
        const int axis = split_axis(s, e);
        const int k = split_index(s, e, axis);
        const int l = build(s, k), r = build(k, e);
        nodes[idx].left = l; nodes[idx].right = r;
        return idx;
    }
};

// One mesh object: the reference's tree over triangles [t0, t0 + n) of `src`; appends them to `out` in visiting order and this repo's
// nodes to `bn` (the reference's tree cut off where a subtree holds <= 4 triangles, children in visiting order).  Returns the root.
int build_part(const std::vector<MeshTri>& src, int t0, int n, std::vector<MeshTri>& out, std::vector<BuildNode>& bn, std::vector<int>* visit_order, std::vector<float>* frames) {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = t0; i < t0 + n; i++) { float mn[3], mx[3]; tri_bounds(src[i], mn, mx); for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], mn[a]); hi[a] = std::max(hi[a], mx[a]); } }
    QBvhFrame F; F.set(lo, hi);
    if (frames) for (int a = 0; a < 9; a++) frames->push_back(a < 3 ? F.mn[a] : (a < 6 ? F.mx[a - 3] : F.q[a - 6]));   // the object's quantization frame (MESH_FRAME_WORDS)
    std::vector<QLeaf> L(n);
    for (int i = 0; i < n; i++) {
        float mn[3], mx[3]; tri_bounds(src[t0 + i], mn, mx);
        for (int a = 0; a < 3; a++) if (mx[a] - mn[a] < 0.002f) { mx[a] = mx[a] + 0.001f; mn[a] = mn[a] - 0.001f; }
        F.quantize(L[i].mn, mn, false); F.quantize(L[i].mx, mx, true); L[i].tri = t0 + i;
    }
    QTree T{F, L, {}};
    T.nodes.reserve(2 * n);
    T.build(0, n);
    // visiting order + this repo's nodes
    struct Rec {
        const QTree& T; const std::vector<MeshTri>& src; std::vector<MeshTri>& out; std::vector<BuildNode>& bn; std::vector<int>* vo;
        void emit(int ti) { const QTree::Node& nd = T.nodes[ti]; if (nd.left < 0) { out.push_back(src[T.L[nd.first].tri]); if (vo) vo->push_back(T.L[nd.first].tri); return; } emit(nd.left); emit(nd.right); }
        int go(int ti) {
            const QTree::Node& nd = T.nodes[ti];
            const int idx = (int)bn.size();
            bn.push_back(BuildNode{});
            if (nd.count <= 4) {   // (<= 64 leaves: no header inside, plain depth-first order)
                bn[idx].first = (int)out.size(); bn[idx].count = nd.count;
                emit(ti);
            } else {
                int a = nd.left, b = nd.right;
                if (T.nodes[a].count <= 64 && T.nodes[b].count > 64) std::swap(a, b);   // the right child's headers were made first
                const int l = go(a), r = go(b);
                bn[idx].left = l; bn[idx].right = r;
            }
            return idx;
        }
    } rec{T, src, out, bn, visit_order};
    return rec.go(0);
}

void fit_bounds(const std::vector<MeshTri>& tris, std::vector<BuildNode>& bn, int i) {
    BuildNode& nd = bn[i];
    for (int a = 0; a < 3; a++) { nd.mn[a] = 1e30f; nd.mx[a] = -1e30f; }
    if (nd.count > 0) {
        for (int t = nd.first; t < nd.first + nd.count; t++) {
            float mn[3], mx[3]; tri_bounds(tris[t], mn, mx);
            for (int a = 0; a < 3; a++) { nd.mn[a] = std::min(nd.mn[a], mn[a]); nd.mx[a] = std::max(nd.mx[a], mx[a]); }
        }
        return;
    }
    fit_bounds(tris, bn, nd.left); fit_bounds(tris, bn, nd.right);
    for (int a = 0; a < 3; a++) { bn[i].mn[a] = std::min(bn[bn[i].left].mn[a], bn[bn[i].right].mn[a]); bn[i].mx[a] = std::max(bn[bn[i].left].mx[a], bn[bn[i].right].mx[a]); }
}

using Key = std::tuple<int64_t, int64_t, int64_t>;
Key qkey(float x, float y, float z) {
    const float q = 1024.f;  // BT units -> 1/1024 BT (~0.05 uu) grid
    return Key{(int64_t)std::llround(x * q), (int64_t)std::llround(y * q), (int64_t)std::llround(z * q)};
}
}  // namespace

HostMesh build_host_mesh(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris, const std::vector<int>* part_tris, bool verts_in_bt) {
    HostMesh m;
    m.tris.resize(n_tris);
    for (int i = 0; i < n_tris; i++) {
        MeshTri& t = m.tris[i];
        float* p = &t.v0x;
        for (int k = 0; k < 3; k++) {
            int vi = tris[i * 3 + k];
            if (vi < 0 || vi >= n_verts) vi = 0;
            for (int a = 0; a < 3; a++) p[k * 3 + a] = verts_in_bt ? verts_uu[vi * 3 + a] : verts_uu[vi * 3 + a] * UU2BT;   // (.cmf files hold Bullet units: kept to the bit)
        }
        t.edge_flags = 0; t.obj = 0; t._pad1 = 0; t._pad2 = 0;
        t.edge_angle[0] = t.edge_angle[1] = t.edge_angle[2] = 6.283185307179586232f;   // btTriangleInfo(): SIMD_2_PI
    }
    // The tree first: per mesh object the reference's own (see build_part), the objects joined pairwise in file order above them.  Triangles
    // are stored in the reference's visiting order from here on, so "ascending first-triangle index" IS that order for any set of leaves
    // -- and the edge records below are made in it too.
    std::vector<BuildNode> bn;
    bn.reserve(2 * n_tris + 64);
    int root = -1;
    std::vector<int> part_n, part_of(n_tris, 0), part_first;
    std::vector<float> frames;   // per mesh object: its btOptimizedBvh's quantization frame (bvhAabbMin, bvhAabbMax, bvhQuantization)
    if (n_tris > 0) {
        std::vector<MeshTri> ordered; ordered.reserve(n_tris);
        std::vector<int> src_of; src_of.reserve(n_tris);
        if (part_tris && !part_tris->empty()) { int sum = 0; for (int c : *part_tris) { if (c > 0) part_n.push_back(c); sum += c; } if (sum != n_tris) part_n.assign(1, n_tris); }
        else part_n.assign(1, n_tris);
        std::vector<int> roots;
        int t0 = 0;
        for (size_t k = 0; k < part_n.size(); k++) {
            const int c = part_n[k];
            part_first.push_back(t0);
            for (int i = t0; i < t0 + c; i++) part_of[i] = (int)k;
            roots.push_back(build_part(m.tris, t0, c, ordered, bn, &src_of, &frames)); t0 += c;
        }
        while (roots.size() > 1) {
            std::vector<int> up;
            for (size_t i = 0; i + 1 < roots.size(); i += 2) { BuildNode j; j.left = roots[i]; j.right = roots[i + 1]; up.push_back((int)bn.size()); bn.push_back(j); }
            if (roots.size() & 1) up.push_back(roots.back());
            roots.swap(up);
        }
        root = roots[0];
        m.tris.swap(ordered);
        for (int i = 0; i < n_tris; i++) m.tris[i].obj = (uint32_t)part_of[i];     // (a part keeps its index range through the reordering)
        m.source_tri.assign(src_of.begin(), src_of.end());
        fit_bounds(m.tris, bn, root);
    }
    // btGenerateInternalEdgeInfo (btInternalEdgeUtility.cpp:295-352) with btConnectivityProcessor::processTriangle (:50-290): for every
    // triangle A, every other triangle B of the same mesh object whose box overlaps A's, in the order processAllTriangles reports them
    // (= stored order); two shared vertices (closer than 1e-4) make a shared edge, whose dihedral angle, convexity and normal-swap flag
    // go into A's record.  Where three triangles meet in one edge the flags of both neighbours accumulate and the LAST neighbour's angle
    // stays, so the order matters (the goal frame of the procedural arena has such edges; tools/edge_fuzz.py found them).
    {
        using namespace std;
        auto V = [&](int tri, int k) { const float* p = &m.tris[tri].v0x; return v3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2]); };
        const float EQ = 0.0001f * 0.0001f, PLANAR = 0.0001f;
        vector<array<float, 6>> box(n_tris);
        for (int i = 0; i < n_tris; i++) { float mn[3], mx[3]; tri_bounds(m.tris[i], mn, mx); box[i] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]}; }
        for (int ia = 0; ia < n_tris; ia++) {
            const V3 A[3] = {V(ia, 0), V(ia, 1), V(ia, 2)};
            const int pf = part_first[part_of[ia]], pl = pf + part_n[part_of[ia]];
            for (int ib = pf; ib < pl; ib++) {
                if (ib == ia) continue;
                bool ov = true;
                for (int a = 0; a < 3; a++) if (box[ia][a] > box[ib][3 + a] || box[ia][3 + a] < box[ib][a]) ov = false;
                if (!ov) continue;
                const V3 B[3] = {V(ib, 0), V(ib, 1), V(ib, 2)};
                if (len2(cross(B[1] - B[0], B[2] - B[0])) < EQ) continue;
                if (len2(cross(A[1] - A[0], A[2] - A[0])) < EQ) continue;
                int ns = 0, sa[3] = {-1, -1, -1}, sb[3] = {-1, -1, -1}; bool degenerate = false;
                for (int i = 0; i < 3 && !degenerate; i++)
                    for (int j = 0; j < 3; j++)
                        if (len2(A[i] - B[j]) < EQ) { sa[ns] = i; sb[ns] = j; ns++; if (ns >= 3) { degenerate = true; break; } }
                if (degenerate || ns != 2) continue;
                if (sa[0] == 0 && sa[1] == 2) { sa[0] = 2; sa[1] = 0; int tmp = sb[1]; sb[1] = sb[0]; sb[0] = tmp; }
                MeshTri& info = m.tris[ia];
                info.edge_flags |= 0x80000000u;
                const int sum = sa[0] + sa[1], other_a = 3 - sum, other_b = 3 - (sb[0] + sb[1]);
                V3 edge = normalized(A[sa[1]] - A[sa[0]]);
                V3 normal_a = normalized(cross(A[1] - A[0], A[2] - A[0]));
                const V3 tb0 = B[sb[1]], tb1 = B[sb[0]], tb2 = B[other_b];
                V3 normal_b = normalized(cross(tb1 - tb0, tb2 - tb0));
                V3 eca = normalized(cross(edge, normal_a));
                if (dot(eca, A[other_a] - A[sa[0]]) < 0) eca *= -1.f;
                V3 ecb = normalized(cross(edge, normal_b));
                if (dot(ecb, B[other_b] - B[sb[0]]) < 0) ecb *= -1.f;
                float corrected = 0.f; bool convex = false;
                V3 ce = cross(eca, ecb);
                if (!(len2(ce) < PLANAR)) {
                    ce = normalized(ce);
                    V3 cna = normalized(cross(ce, eca));
                    float angle2 = atan2f(dot(ecb, cna), dot(ecb, eca));     // btGetAngle(calculatedNormalA, edgeCrossA, edgeCrossB)
                    float ang4 = 3.1415926535897931160f - angle2;
                    convex = dot(normal_a, ecb) < 0.f;
                    corrected = convex ? ang4 : -ang4;
                }
                // which edge of A, its rotation axis, and whether the neighbour's normal comes out flipped
                V3 axis; int e;
                if (sum == 1) { axis = A[0] - A[1]; e = 0; } else if (sum == 2) { axis = A[2] - A[0]; e = 2; } else { axis = A[1] - A[2]; e = 1; }
                Q4 orn = quat_axis_angle(axis, -corrected);
                V3 cnb = quat_rotate(orn, normal_a);
                const uint32_t bit_convex = e == 0 ? 1u : (e == 1 ? 2u : 4u), bit_swap = e == 0 ? 8u : (e == 1 ? 16u : 32u);
                if (dot(cnb, normal_b) < 0) info.edge_flags |= bit_swap;
                info.edge_angle[e] = -corrected;
                if (convex) info.edge_flags |= bit_convex;
            }
        }
    }
    // occupancy grid: a cell is marked when a triangle's AABB and its supporting plane both cut the cell box grown by a
    // small margin (conservative superset of exact triangle/box overlap; tight for the axis-aligned and 45-degree walls)
    m.grid.assign(GRID_WORDS, 0u);
    for (int i = 0; i < n_tris; i++) {
        const float* p = &m.tris[i].v0x;
        float mn[3], mx[3]; tri_bounds(m.tris[i], mn, mx);
        P3 v0{p[0], p[1], p[2]}, v1{p[3], p[4], p[5]}, v2{p[6], p[7], p[8]};
        P3 n = crs(sub(v1, v0), sub(v2, v0));
        float nl = std::sqrt(dt3(n, n));
        const float gmin[3] = {GRID_MIN_X, GRID_MIN_Y, GRID_MIN_Z}; const int gdim[3] = {GRID_X, GRID_Y, GRID_Z};
        int c0[3], c1[3];
        const float margin = 0.02f;
        for (int a = 0; a < 3; a++) {
            c0[a] = std::max(0, (int)std::floor((mn[a] - margin - gmin[a]) / GRID_CELL));
            c1[a] = std::min(gdim[a] - 1, (int)std::floor((mx[a] + margin - gmin[a]) / GRID_CELL));
        }
        for (int z = c0[2]; z <= c1[2]; z++) for (int y = c0[1]; y <= c1[1]; y++) for (int x = c0[0]; x <= c1[0]; x++) {
            bool cut = true;
            if (nl > 1e-12f) {
                P3 nn{n.x / nl, n.y / nl, n.z / nl};
                float cx = gmin[0] + (x + 0.5f) * GRID_CELL, cy = gmin[1] + (y + 0.5f) * GRID_CELL, cz = gmin[2] + (z + 0.5f) * GRID_CELL;
                float dist = nn.x * (cx - v0.x) + nn.y * (cy - v0.y) + nn.z * (cz - v0.z);
                float rad = (GRID_CELL * 0.5f + margin) * (std::fabs(nn.x) + std::fabs(nn.y) + std::fabs(nn.z));
                cut = std::fabs(dist) <= rad;
            }
            if (cut) { int bit = (z * GRID_Y + y) * GRID_X + x; m.grid[bit >> 5] |= (1u << (bit & 31)); }
        }
    }
    // The reference's broadphase grid (btRSBroadphase.cpp:95-182; cell and bounds in arena_contact.h): a dynamic body is paired with a
    // mesh OBJECT (one per .cmf file: Arena.cpp:1028-1054) iff the object is on the static list of the body's cell, i.e. iff some cell of
    // its 27-neighbourhood holds one of the object's triangles.  Appended to the occupancy words: the number of objects, the objects' own
    // boxes (6 floats each) for the pair's AABB test, and per broadphase cell the mask of the objects listed there.
    {
        const int NXYZ = BP_CELLS_X * BP_CELLS_Y * BP_CELLS_Z;
        const float mnp[3] = {-4500.f * UU2BT, -6000.f * UU2BT, 0.f * UU2BT}; const int dim[3] = {BP_CELLS_X, BP_CELLS_Y, BP_CELLS_Z};
        int n_obj = 0;
        for (int i = 0; i < n_tris; i++) n_obj = std::max(n_obj, (int)m.tris[i].obj + 1);
        if (n_obj == 0) n_obj = 1;
        if (n_obj > BP_MAX_OBJECTS) n_obj = BP_MAX_OBJECTS;     // (objects beyond the mask's width share the last bit: only the ORDER of empty manifolds is affected)
        // layout: [n_obj] [n_obj x 6 floats: the objects' boxes] [one word per broadphase cell: bit o = object o is listed there]
        m.grid.resize(GRID_WORDS + 1 + (size_t)n_obj * 6 + (size_t)NXYZ, 0u);
        m.grid[GRID_WORDS] = (uint32_t)n_obj;
        uint32_t* boxes = &m.grid[GRID_WORDS + 1];
        uint32_t* cellmask = &m.grid[GRID_WORDS + 1 + (size_t)n_obj * 6];
        for (int o = 0; o < n_obj; o++) {
            std::vector<uint8_t> has(NXYZ, 0);
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (int i = 0; i < n_tris; i++) {
                if (std::min((int)m.tris[i].obj, BP_MAX_OBJECTS - 1) != o) continue;
                float mn[3], mx[3]; tri_bounds(m.tris[i], mn, mx);
                int c0[3], c1[3];
                for (int a = 0; a < 3; a++) {
                    lo[a] = std::min(lo[a], mn[a]); hi[a] = std::max(hi[a], mx[a]);
                    c0[a] = std::max(0, (int)std::floor((mn[a] - mnp[a]) / BP_CELL) - 1);
                    c1[a] = std::min(dim[a] - 1, (int)std::floor((mx[a] - mnp[a]) / BP_CELL) + 1);
                }
                for (int x = c0[0]; x <= c1[0]; x++) for (int y = c0[1]; y <= c1[1]; y++) for (int z = c0[2]; z <= c1[2]; z++) {
                    const float cmn[3] = {mnp[0] + x * BP_CELL, mnp[1] + y * BP_CELL, mnp[2] + z * BP_CELL};
                    bool ov = true;
                    for (int a = 0; a < 3; a++) if (mn[a] > cmn[a] + BP_CELL || mx[a] < cmn[a]) ov = false;
                    if (ov) has[bp_cell_index(x, y, z)] = 1;
                }
            }
            for (int x = 0; x < dim[0]; x++) for (int y = 0; y < dim[1]; y++) for (int z = 0; z < dim[2]; z++) {
                bool listed = false;
                for (int dx = -1; dx <= 1 && !listed; dx++) for (int dy = -1; dy <= 1 && !listed; dy++) for (int dz = -1; dz <= 1 && !listed; dz++) {
                    int xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || yy < 0 || zz < 0 || xx >= dim[0] || yy >= dim[1] || zz >= dim[2]) continue;
                    listed = has[bp_cell_index(xx, yy, zz)] != 0;
                }
                if (listed) cellmask[bp_cell_index(x, y, z)] |= 1u << o;
            }
            for (int a = 0; a < 3; a++) { memcpy(&boxes[o * 6 + a], &lo[a], 4); memcpy(&boxes[o * 6 + 3 + a], &hi[a], 4); }
        }
        // ... and behind the cell masks, per mesh object (ALL of them, by MeshTri::obj) the quantization frame of its btOptimizedBvh: what a ray needs to know
        // whether Bullet's tree walk would have reached a triangle's leaf at all (arena_world.h ray_leaf_admits)
        const size_t at = m.grid.size();
        m.grid.resize(at + frames.size(), 0u);
        if (!frames.empty()) memcpy(&m.grid[at], frames.data(), 4 * frames.size());
    }
    if (n_tris == 0) return m;
    // breadth-first renumbering with sibling pairs adjacent
    std::vector<int> order; order.reserve(bn.size());
    std::vector<int> newidx(bn.size(), -1);
    std::queue<int> q; q.push(root);
    order.push_back(root); newidx[root] = 0;
    while (!q.empty()) {
        int i = q.front(); q.pop();
        if (bn[i].count == 0) {
            newidx[bn[i].left] = (int)order.size(); order.push_back(bn[i].left);
            newidx[bn[i].right] = (int)order.size(); order.push_back(bn[i].right);
            q.push(bn[i].left); q.push(bn[i].right);
        }
    }
    m.nodes.resize(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        const BuildNode& b = bn[order[k]];
        BvhNode& n = m.nodes[k];
        n.minx = b.mn[0]; n.miny = b.mn[1]; n.minz = b.mn[2]; n.maxx = b.mx[0]; n.maxy = b.mx[1]; n.maxz = b.mx[2];
        if (b.count > 0) { n.left_or_first = b.first; n.count_escape = (uint32_t)b.count; }
        else { n.left_or_first = newidx[b.left]; n.count_escape = 0; }
    }
    // thread the tree for the stackless depth-first walk, left child first (= visiting order): after the left subtree comes the right
    // sibling, after the right subtree whatever follows the parent
    {
        std::vector<std::pair<int, uint32_t>> todo; todo.push_back({0, BVH_END});
        while (!todo.empty()) {
            auto [k, esc] = todo.back(); todo.pop_back();
            BvhNode& n = m.nodes[k];
            n.count_escape = (n.count_escape & 0xffu) | (esc << 8);
            if ((n.count_escape & 0xffu) == 0) {
                int left = n.left_or_first;
                todo.push_back({left, (uint32_t)(left + 1)});
                todo.push_back({left + 1, esc});
            }
        }
    }
    return m;
}

namespace {
struct MeshOut {
    std::vector<float>& v; std::vector<int32_t>& t;
    int vert(float x, float y, float z) { v.push_back(x); v.push_back(y); v.push_back(z); return (int)(v.size() / 3) - 1; }
    void quad(const float a[3], const float b[3], const float c[3], const float d[3]) {
        int i0 = vert(a[0], a[1], a[2]), i1 = vert(b[0], b[1], b[2]), i2 = vert(c[0], c[1], c[2]), i3 = vert(d[0], d[1], d[2]);
        t.push_back(i0); t.push_back(i1); t.push_back(i2);
        t.push_back(i0); t.push_back(i2); t.push_back(i3);
    }
    // quarter-cylinder fillet along the segment p0->p1 lying in the junction of a horizontal surface (floor or
    // ceiling, `up` = +1 / -1) and a vertical wall whose inward horizontal unit normal is (inx, iny)
    void fillet(float x0, float y0, float x1, float y1, float zbase, float up, float inx, float iny, float R, int segs) {
        for (int s = 0; s < segs; s++) {
            float a0 = (float)(M_PI / 2) * s / segs, a1 = (float)(M_PI / 2) * (s + 1) / segs;
            auto pt = [&](float x, float y, float a, float* o) {
                float cx = x + inx * R, cy = y + iny * R, cz = zbase + up * R;
                o[0] = cx - inx * R * std::sin(a); o[1] = cy - iny * R * std::sin(a); o[2] = cz - up * R * std::cos(a);
            };
            float A[3], B[3], C[3], D[3];
            pt(x0, y0, a0, A); pt(x1, y1, a0, B); pt(x1, y1, a1, C); pt(x0, y0, a1, D);
            quad(A, B, C, D);
        }
    }
};
}  // namespace

void make_procedural_soccar(std::vector<float>& verts, std::vector<int32_t>& tris) { make_procedural_soccar_ex(verts, tris, 4, 0.f); }

// Every edge longer than max_edge is split at its midpoint (a triangle with 1 / 2 / 3 such edges becomes 2 / 3 / 4), until none is left.
// The decision is per EDGE, so neighbours split the edge they share at the same point and the surface stays free of T-junctions.
static void subdivide_long_edges(std::vector<float>& v, std::vector<int32_t>& t, float max_edge) {
    if (!(max_edge > 0.f)) return;
    const float lim2 = max_edge * max_edge;
    auto P = [&](int i) { return &v[3 * (size_t)i]; };
    auto len2 = [&](int a, int b) { const float* p = P(a); const float* q = P(b); const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2]; return dx * dx + dy * dy + dz * dz; };
    auto mid = [&](int a, int b) { const float x = (P(a)[0] + P(b)[0]) * 0.5f, y = (P(a)[1] + P(b)[1]) * 0.5f, z = (P(a)[2] + P(b)[2]) * 0.5f; v.push_back(x); v.push_back(y); v.push_back(z); return (int)(v.size() / 3) - 1; };
    for (bool any = true; any;) {
        any = false;
        std::vector<int32_t> out; out.reserve(t.size() * 2);
        for (size_t k = 0; k + 2 < t.size(); k += 3) {
            const int a = t[k], b = t[k + 1], c = t[k + 2];
            const bool sab = len2(a, b) > lim2, sbc = len2(b, c) > lim2, sca = len2(c, a) > lim2;
            const int n = (int)sab + (int)sbc + (int)sca;
            auto tri = [&](int x, int y, int z) { out.push_back(x); out.push_back(y); out.push_back(z); };
            if (n == 0) { tri(a, b, c); continue; }
            any = true;
            const int mab = sab ? mid(a, b) : -1, mbc = sbc ? mid(b, c) : -1, mca = sca ? mid(c, a) : -1;
            if (n == 3) { tri(a, mab, mca); tri(mab, b, mbc); tri(mca, mbc, c); tri(mab, mbc, mca); }
            else if (n == 1) { if (sab) { tri(a, mab, c); tri(mab, b, c); } else if (sbc) { tri(a, b, mbc); tri(a, mbc, c); } else { tri(a, b, mca); tri(mca, b, c); } }
            else if (!sab) { tri(a, b, mbc); tri(a, mbc, mca); tri(mca, mbc, c); }     // bc and ca split
            else if (!sbc) { tri(a, mab, mca); tri(mab, b, c); tri(mab, c, mca); }     // ab and ca split
            else { tri(a, mab, mbc); tri(a, mbc, c); tri(mab, b, mbc); }               // ab and bc split
        }
        t.swap(out);
    }
}

void make_procedural_soccar_ex(std::vector<float>& verts, std::vector<int32_t>& tris, int fillet_segments, float max_edge_uu) {
    MeshOut m{verts, tris};
    const float X = 4096.f, Y = 5120.f, Z = 2048.f, GX = 892.755f, GZ = 642.775f, GY = 6000.f, CUT = 1152.f, R = 256.f;
    const int SEG = fillet_segments > 0 ? fillet_segments : 4;
    for (int s = -1; s <= 1; s += 2) {
        float y = s * Y, yb = s * GY;
        // back wall around the goal mouth
        { float a[3] = {-(X - CUT), y, 0}, b[3] = {-GX, y, 0}, c[3] = {-GX, y, Z}, d[3] = {-(X - CUT), y, Z}; m.quad(a, b, c, d); }
        { float a[3] = {GX, y, 0}, b[3] = {X - CUT, y, 0}, c[3] = {X - CUT, y, Z}, d[3] = {GX, y, Z}; m.quad(a, b, c, d); }
        { float a[3] = {-GX, y, GZ}, b[3] = {GX, y, GZ}, c[3] = {GX, y, Z}, d[3] = {-GX, y, Z}; m.quad(a, b, c, d); }
        // goal box: sides, back, roof
        { float a[3] = {-GX, y, 0}, b[3] = {-GX, yb, 0}, c[3] = {-GX, yb, GZ}, d[3] = {-GX, y, GZ}; m.quad(a, b, c, d); }
        { float a[3] = {GX, y, 0}, b[3] = {GX, yb, 0}, c[3] = {GX, yb, GZ}, d[3] = {GX, y, GZ}; m.quad(a, b, c, d); }
        { float a[3] = {-GX, yb, 0}, b[3] = {GX, yb, 0}, c[3] = {GX, yb, GZ}, d[3] = {-GX, yb, GZ}; m.quad(a, b, c, d); }
        { float a[3] = {-GX, y, GZ}, b[3] = {GX, y, GZ}, c[3] = {GX, yb, GZ}, d[3] = {-GX, yb, GZ}; m.quad(a, b, c, d); }
        // corner walls (45 degrees) on both x sides
        for (int sx = -1; sx <= 1; sx += 2) {
            float a[3] = {sx * (X - CUT), y, 0}, b[3] = {sx * X, s * (Y - CUT), 0}, c[3] = {sx * X, s * (Y - CUT), Z}, d[3] = {sx * (X - CUT), y, Z};
            m.quad(a, b, c, d);
            const float inv = 0.70710678f;
            for (int up = -1; up <= 1; up += 2)
                m.fillet(sx * (X - CUT), y, sx * X, s * (Y - CUT), up > 0 ? 0.f : Z, (float)up, -sx * inv, -s * inv, R, SEG);
        }
        // fillets along the back wall (left and right of the goal on the floor, full width on the ceiling)
        m.fillet(-(X - CUT), y, -GX, y, 0.f, 1.f, 0.f, (float)-s, R, SEG);
        m.fillet(GX, y, X - CUT, y, 0.f, 1.f, 0.f, (float)-s, R, SEG);
        m.fillet(-(X - CUT), y, X - CUT, y, Z, -1.f, 0.f, (float)-s, R, SEG);
    }
    // fillets along the side walls (the walls themselves are planes, Arena.cpp:1060-1101)
    for (int sx = -1; sx <= 1; sx += 2)
        for (int up = -1; up <= 1; up += 2)
            m.fillet(sx * X, -(Y - CUT), sx * X, (Y - CUT), up > 0 ? 0.f : Z, (float)up, (float)-sx, 0.f, R, SEG);
    subdivide_long_edges(verts, tris, max_edge_uu);
}

bool append_cmf(const uint8_t* data, size_t size, std::vector<float>& verts, std::vector<int32_t>& tris, bool keep_bt) {
    if (size < 8) return false;
    int32_t nt, nv;
    memcpy(&nt, data, 4); memcpy(&nv, data + 4, 4);
    if (nt < 0 || nv < 0 || size < 8 + (size_t)nt * 12 + (size_t)nv * 12) return false;
    int base = (int)(verts.size() / 3);
    const uint8_t* p = data + 8;
    for (int i = 0; i < nt * 3; i++) { int32_t v; memcpy(&v, p, 4); p += 4; tris.push_back(base + v); }
    for (int i = 0; i < nv * 3; i++) { float v; memcpy(&v, p, 4); p += 4; verts.push_back(keep_bt ? v : v * BT2UU); }
    return true;
}

}  // namespace rlg
