// Development helper (host only): prints where the plan of csrc/cg_big.hpp puts every array for a given size.
//   hipcc -std=c++17 --offload-arch=gfx950 -o tools/plan_dump/plan_dump tools/plan_dump/plan_dump.hip && tools/plan_dump/plan_dump n nthr cap_kb [mode]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include "../../coulombgas_amd/csrc/cg_big.hpp"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 57, nt = argc > 2 ? atoi(argv[2]) : 512, kb = argc > 3 ? atoi(argv[3]) : 159, mode = argc > 4 ? atoi(argv[4]) : 2;
    const size_t cap = (size_t)kb * 1024 / 8 - CG_TAB_DOUBLES;
    using Bg = CgBig<2, 16, 16>;
    auto s = Bg::layout_scores(n, nt, cap);
    printf("scores  n=%d nt=%d cap %zu doubles: ok=%d lds %u ws %u\n", n, nt, cap, s.ok, s.lds_total, s.ws_total);
    auto g = Bg::layout_gradlap(n, nt, mode, cap);
    printf("gradlap n=%d nt=%d mode %d: ok=%d lds %u ws %u\n", n, nt, mode, g.ok, g.lds_total, g.ws_total);
    return 0;
}
