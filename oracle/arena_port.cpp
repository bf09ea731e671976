// oracle/arena_port.cpp — TEST INFRASTRUCTURE ONLY.
//
// Host (g++) build of the arena stepper's scalar core.  The physics tick is ~3k lines of branchy fp32 code
// that has to be developed and debugged in a container without a GPU, so the tick is written ONCE in
// rlgymppo_cpp_amd/csrc/arena_*.h as host/device-neutral inline functions; this file compiles those headers
// for the CPU and exposes them to the tests.  It is the "port" CPU restatement of the stepper:
//   * pinned against the REAL reference (oracle/_ref, golden trajectories under tests/golden/) by
//     tests/test_arena_port_vs_ref.py, and
//   * used by the -m gpu tests as the tick-by-tick oracle of the HIP kernel (same source, different
//     compiler/ISA: agreement is to fp32 rounding of libm calls).
// The dependency runs oracle -> product headers only; nothing in rlgymppo_cpp_amd/ includes, links or calls
// anything in oracle/ (the product path fails loudly without its HIP library).
#include "../rlgymppo_cpp_amd/csrc/arena_gym.h"
#include "../rlgymppo_cpp_amd/csrc/arena_mesh.h"
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

using namespace rlg;

namespace {
HostMesh g_mesh;
MeshView view() {
    MeshView v; v.nodes = g_mesh.nodes.data(); v.tris = g_mesh.tris.data(); v.nodes_fast = nullptr;
    v.n_nodes = (int)g_mesh.nodes.size(); v.n_tris = (int)g_mesh.tris.size(); v.n_fast = 0;
    return v;
}
template <int NC>
void step_t(RlgpuArenaState* s, int ticks, uint32_t seed, uint32_t env) {
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, *s);
    MeshView mv = view();
    for (int t = 0; t < ticks; t++) { TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, seed, env, ev); }
    arena_to_host(A, G, *s);
}
}  // namespace

extern "C" {

int port_state_size() { return (int)sizeof(RlgpuArenaState); }

void port_set_mesh(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris) {
    g_mesh = build_host_mesh(verts_uu, n_verts, tris, n_tris);
}
int port_procedural_mesh(float* verts_out, int cap_verts, int32_t* tris_out, int cap_tris, int* n_verts, int* n_tris) {
    std::vector<float> v; std::vector<int32_t> t;
    make_procedural_soccar(v, t);
    *n_verts = (int)v.size() / 3; *n_tris = (int)t.size() / 3;
    if (*n_verts > cap_verts || *n_tris > cap_tris) return -1;
    memcpy(verts_out, v.data(), v.size() * 4); memcpy(tris_out, t.data(), t.size() * 4);
    return 0;
}
int port_mesh_counts(int* n_nodes, int* n_tris) { *n_nodes = (int)g_mesh.nodes.size(); *n_tris = (int)g_mesh.tris.size(); return 0; }

// advance the physical state by `ticks` ticks with the controls stored in the state
void port_arena_step(RlgpuArenaState* s, int ticks, uint32_t seed, uint32_t env) {
    if (s->num_cars == 2) step_t<2>(s, ticks, seed, env);
    else if (s->num_cars == 4) step_t<4>(s, ticks, seed, env);
    else if (s->num_cars == 6) step_t<6>(s, ticks, seed, env);
}

}  // extern "C"
