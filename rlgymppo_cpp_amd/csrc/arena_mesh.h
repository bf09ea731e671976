// arena_mesh.h — host-side construction of the static arena mesh: the procedural soccar stand-in,
// the .cmf reader (CollisionMeshFile.cpp:11-36 format), per-edge flags and this repo's BVH.
//
// The reference shares one quantized btBvhTriangleMeshShape per .cmf file between all arenas
// (RocketSim.cpp:149-167, Arena.cpp:1054-1057).  Here all meshes are merged into one triangle soup with one
// binary AABB tree laid out breadth-first, so the first n nodes (the top levels) are contiguous and can be
// staged in LDS by the stepper kernel.  The SET of triangles a query reports and the ORDER they are collided in
// are the reference's: per mesh object the tree is btOptimizedBvh's own (arena_mesh.cpp:build_part restates its
// build), cut off at <= 4 triangles per leaf, and the triangles are stored in its visiting order.
#pragma once
#include <vector>
#include <cstdint>
#include "arena_types.h"

namespace rlg {

struct HostMesh {
    std::vector<MeshTri> tris;   // BT units, in the reference's visiting order (object by object)
    std::vector<int32_t> source_tri;   // tris[i] is triangle source_tri[i] of the input
    std::vector<BvhNode> nodes;  // breadth-first
    std::vector<uint32_t> grid;  // GRID_WORDS occupancy bits (arena_types.h)
};

// verts in uu, tris index triplets; part_tris: triangles per mesh object (.cmf file) in input order, or null = one object
// (verts_in_bt: the vertices are already in Bullet units, as .cmf files hold them -- no uu round trip, the triangles are the reference's to the bit)
HostMesh build_host_mesh(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris, const std::vector<int>* part_tris = nullptr, bool verts_in_bt = false);

// the procedural soccar arena (uu): back walls with goal mouths, goal boxes, 45-degree corner walls and
// quarter-cylinder floor fillets.  Geometry facts from RLConst.h:14-16,109, Arena.cpp:846-849, CommonValues.h:9-13.
void make_procedural_soccar(std::vector<float>& verts_uu, std::vector<int32_t>& tris);
// the same arena at a chosen resolution: quarter-cylinder fillets of `fillet_segments` strips, every edge longer than max_edge_uu split
// (0 = none) -- a stand-in for the triangle counts of the game's own soccar meshes (thousands of triangles in 16 files)
void make_procedural_soccar_ex(std::vector<float>& verts_uu, std::vector<int32_t>& tris, int fillet_segments, float max_edge_uu);

// parse one .cmf blob (i32 nTris, i32 nVerts, tris, verts in BT units) and append to verts(uu)/tris
bool append_cmf(const uint8_t* data, size_t size, std::vector<float>& verts_uu, std::vector<int32_t>& tris, bool keep_bt = false);

}  // namespace rlg
