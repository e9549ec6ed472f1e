// tests/tools/sanitize_driver.cpp — native driver for tests/tools/sanitize_cpu.sh: parses the scene files given on the command line
// with the ASan/UBSan build of the loader (error paths included: they throw C++ exceptions inside the library, which the
// LD_PRELOADed-ASan Python run cannot host).
#include <cstdio>

#include "../../include/radish_host.h"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
        rdh_parsed_scene *s = nullptr;
        char err[512];
        int rc = rdh_scene_parse(argv[i], nullptr, nullptr, &s, err, sizeof(err));
        std::printf("%s -> rc %d (%s), prims %d, textures %d\n", argv[i], rc, rc ? err : "ok", s ? s->numPrims : -1, s ? s->numTextures : -1);
        rdh_scene_parse_free(s);
    }
    return 0;
}
