// Host-side scene build on the CPU (no HIP): phase times of build_host_scene with RAYCA_BUILD_TIMING=1.
// A tool for working on the builder without a GPU; built and driven by tests/host_build_probe.py.
#include "../../rayca_amd/csrc/host_scene.hpp"
#include <chrono>
#include <cstdio>
extern "C" int host_build_probe(const RaycaSceneDesc* d, unsigned builder, int repeats) {
  for (int r = 0; r < repeats; ++r) {
    rayca::HostScene s;
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t rc = rayca::build_host_scene(*d, true, builder, s, err);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc) { fprintf(stderr, "build failed: %s\n", err.c_str()); return rc; }
    fprintf(stderr, "[probe] build #%d: %.1f ms, %zu nodes, %zu 4-wide\n", r, std::chrono::duration<float, std::milli>(t1 - t0).count(), s.dev_nodes.size(), s.dev_nodes4.size());
  }
  return 0;
}
