// TEST INFRASTRUCTURE.  Hostile inputs for the host layer's two file parsers (raytracing_amd/csrc/host:
// rt_decode_jpeg, rt_load_model), built with -fsanitize=address,undefined by `make -C oracle asan` and driven by
// tests/test_host_hostile_inputs.py.  The reference's own counterparts are stb_image and the model loader at
// src/SceneObjects.cpp:211-255, which trusts its header and its indices.
//
//   host_fuzz jpeg FILE N_CUTS N_FLIPS SEED   the file cut short at N_CUTS places, then N_FLIPS copies with a
//                                             few bytes changed: each through rt_decode_jpeg
//   host_fuzz model FILE...                   each file through rt_load_model; indices are then used the way
//                                             the scene builder uses them
// Prints one line per input: "ok W H" / "ok NV NT" or "error: <text>".  The parsers may accept or refuse an input;
// what they may not do is read or write out of bounds, which ends the process with a sanitizer report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_host.h"

static std::vector<unsigned char> readAll(const char *path) {
    std::vector<unsigned char> out;
    FILE *fp = fopen(path, "rb");
    if (!fp) return out;
    unsigned char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) out.insert(out.end(), buf, buf + got);
    fclose(fp);
    return out;
}

static void decode(const std::vector<unsigned char> &bytes, const std::string &tmp) {
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) exit(2);
    if (!bytes.empty()) fwrite(bytes.data(), 1, bytes.size(), fp);
    fclose(fp);
    unsigned char *rgb = nullptr;
    int w = 0, h = 0, n = 0;
    if (rt_decode_jpeg(tmp.c_str(), &rgb, &w, &h, &n) == 0) {
        // every byte of what came back must be there
        unsigned long long sum = 0;
        for (size_t i = 0; i < (size_t)w * h * 3; ++i) sum += rgb[i];
        printf("ok %d %d %llu\n", w, h, sum & 0xff);
        rt_free(rgb);
    } else {
        printf("error: %s\n", rt_host_last_error());
    }
}

int main(int argc, char **argv) {
    if (argc >= 6 && !strcmp(argv[1], "jpeg")) {
        const std::vector<unsigned char> whole = readAll(argv[2]);
        if (whole.empty()) return 2;
        const int cuts = atoi(argv[3]), flips = atoi(argv[4]);
        uint64_t state = strtoull(argv[5], nullptr, 10) * 6364136223846793005ull + 1442695040888963407ull;
        auto next = [&]() {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            return (uint32_t)(state >> 33);
        };
        const std::string tmp = std::string(argv[2]) + ".fuzz.tmp";
        for (int k = 0; k < cuts; ++k) {  // headers, tables, the first scan, deep inside the data, the last bytes
            const size_t at = k == 0 ? 0 : (k == 1 ? 2 : (k == 2 ? 20 : (size_t)((double)whole.size() * k / cuts)));
            decode(std::vector<unsigned char>(whole.begin(), whole.begin() + (at < whole.size() ? at : whole.size() - 1)), tmp);
        }
        for (int k = 0; k < flips; ++k) {
            std::vector<unsigned char> v = whole;
            // half of the copies are damaged in the headers and tables (the first 700 bytes), half anywhere
            const size_t span = (k & 1) ? (whole.size() < 700 ? whole.size() : 700) : whole.size();
            for (int j = 0; j < 1 + (int)(next() % 6); ++j) v[next() % span] = (unsigned char)next();
            decode(v, tmp);
        }
        remove(tmp.c_str());
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "model")) {
        for (int a = 2; a < argc; ++a) {
            float *verts = nullptr;
            int *idx = nullptr, nv = 0, nt = 0;
            const float off[3] = {4.0f, -5.6f, -23.0f};
            if (rt_load_model(argv[a], 45.0f, off, &verts, &nv, &idx, &nt) == 0) {
                double sum = 0;  // every index must name a vertex that is there
                for (int i = 0; i < 3 * nt; ++i) sum += verts[3 * (size_t)idx[i]];
                printf("ok %d %d %g\n", nv, nt, sum);
                rt_free(verts), rt_free(idx);
            } else {
                printf("error: %s\n", rt_host_last_error());
            }
        }
        return 0;
    }
    fprintf(stderr, "usage: host_fuzz jpeg FILE N_CUTS N_FLIPS SEED | host_fuzz model FILE...\n");
    return 2;
}
