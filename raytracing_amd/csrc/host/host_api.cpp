// extern "C" face of the host layer (include/rt_host.h): model loader, BMP writer, command line.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../../include/rt_host.h"
#include "host_error.hpp"
#include "scene.hpp"

namespace {
thread_local std::string g_host_err;
int hostFail(const std::string &msg) {
    g_host_err = msg;
    return -1;
}

}  // namespace

void rt::setHostError(const std::string &msg) { g_host_err = msg; }

namespace {
// command-line errors of the reference (include/ParseException.h, src/ParseException.cpp)
struct ParseException {
    enum Type { NOT_ENOUGH, SAME_ARGS } type;
    void printMessage() const {
        if (type == NOT_ENOUGH)
            std::cerr << "Not enough arguments." << std::endl;
        else
            std::cerr << "Too many values for the same argument." << std::endl;
    }
};

// src/rt.cpp:7-35: no arguments = defaults; otherwise the first three flag/value pairs are read and
// must be present (argc >= 7); a repeated flag is an error; unknown flags among them are skipped.
// Extension: further pairs after the first three set the values the reference fixes at compile time.
rt::Settings parseArgs(int argc, char **argv) {
    rt::Settings s;
    if (argc == 1) return s;
    const int paramCount = 4;
    if (argc < 2 * paramCount - 1) throw ParseException{ParseException::NOT_ENOUGH};
    s.out.clear(), s.scene = 0, s.threads = 0;
    for (int i = 1; i < 2 * paramCount - 1; i += 2) {
        const std::string arg = argv[i];
        if (arg == "-out") {
            if (!s.out.empty()) throw ParseException{ParseException::SAME_ARGS};
            s.out = argv[i + 1];
        } else if (arg == "-scene") {
            if (s.scene != 0) throw ParseException{ParseException::SAME_ARGS};
            s.scene = (unsigned)atoi(argv[i + 1]);
        } else if (arg == "-threads") {
            if (s.threads != 0) throw ParseException{ParseException::SAME_ARGS};
            s.threads = (unsigned)atoi(argv[i + 1]);
        }
    }
    for (int i = 2 * paramCount - 1; i + 1 < argc; i += 2) {
        const std::string arg = argv[i];
        if (arg == "-w") s.width = atoi(argv[i + 1]);
        else if (arg == "-h") s.height = atoi(argv[i + 1]);
        else if (arg == "-depth") s.depth = atoi(argv[i + 1]);
        else if (arg == "-envmap") s.envmap = argv[i + 1];
        else if (arg == "-resources") s.resources = argv[i + 1];
        else if (arg == "-device") s.device = atoi(argv[i + 1]);
        else if (arg == "-gpus") s.gpus = atoi(argv[i + 1]);
    }
    return s;
}
}  // namespace

extern "C" {

const char *rt_host_last_error(void) { return g_host_err.c_str(); }

int rt_default_materials(float *m9) {
    const rt::Materials m;
    const std::vector<float> f = m.flat();
    memcpy(m9, f.data(), f.size() * sizeof(float));
    return (int)m.table().size();
}

int rt_material_index(const char *name) {
    static const rt::Materials m;
    return m.index(name ? name : "");
}

int rt_default_camera(rtx_camera *cam) {
    if (!cam) return hostFail("rt_default_camera: null argument");
    rt::defaultCamera(cam);
    return 0;
}

// Model::Model, src/SceneObjects.cpp:211-255.  Tokens are separated by white space (the asset has
// CRLF line ends); numbers go through strtod exactly like `file >> double`.
int rt_load_model(const char *path, float scale, const float *off, float **verts, int *nv, int **idx, int *nt) {
    if (!path || !off || !verts || !nv || !idx || !nt) return hostFail("rt_load_model: null argument");
    *verts = nullptr, *idx = nullptr, *nv = 0, *nt = 0;
    FILE *fp = fopen(path, "rb");
    if (!fp) return hostFail(std::string("Failed to open '") + path + "' file.");
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
    fclose(fp);
    const char *p = text.c_str();
    auto nextLong = [&](long &v) {
        char *end;
        errno = 0;
        v = strtol(p, &end, 10);
        if (end == p) return false;
        p = end;
        return true;
    };
    auto nextDouble = [&](double &v) {
        char *end;
        v = strtod(p, &end);
        if (end == p) return false;
        p = end;
        return true;
    };
    auto nextChar = [&]() {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
        return *p ? *p++ : '\0';
    };
    long vertNum = 0, faceNum = 0;
    if (!nextLong(vertNum) || !nextLong(faceNum) || vertNum < 0 || faceNum < 0 || vertNum > (1 << 28) || faceNum > (1 << 28))
        return hostFail("rt_load_model: bad header");
    // a record is at least "v 0 0 0": a header that promises more records than the file can hold is not believed
    // (the arrays below are sized by it)
    if ((unsigned long long)(vertNum + faceNum) * 7 > (unsigned long long)text.size() + 7)
        return hostFail("rt_load_model: the header counts more records than the file holds");
    float *v = (float *)malloc(sizeof(float) * 3 * (size_t)(vertNum ? vertNum : 1));
    int *f = (int *)malloc(sizeof(int) * 3 * (size_t)(faceNum ? faceNum : 1));
    if (!v || !f) {
        free(v), free(f);
        return hostFail("rt_load_model: out of memory");
    }
    bool ok = true;
    for (long i = 0; i < vertNum && ok; ++i) {
        double x = 0, y = 0, z = 0;
        nextChar();  // 'v'
        ok = nextDouble(x) && nextDouble(y) && nextDouble(z);
        // vec3((float)x, (float)y, (float)z) * scale + offset
        v[3 * i + 0] = (float)x * scale + off[0];
        v[3 * i + 1] = (float)y * scale + off[1];
        v[3 * i + 2] = (float)z * scale + off[2];
    }
    for (long i = 0; i < faceNum && ok; ++i) {
        long a = 0, b = 0, c = 0;
        nextChar();  // 'f'
        ok = nextLong(a) && nextLong(b) && nextLong(c);
        if (ok && (a < 1 || b < 1 || c < 1 || a > vertNum || b > vertNum || c > vertNum)) ok = false;
        f[3 * i] = (int)a - 1, f[3 * i + 1] = (int)b - 1, f[3 * i + 2] = (int)c - 1;
    }
    if (!ok) {
        free(v), free(f);
        return hostFail("rt_load_model: malformed vertex or face record");
    }
    *verts = v, *nv = (int)vertNum, *idx = f, *nt = (int)faceNum;
    return 0;
}

void rt_free(void *p) { free(p); }

int rt_scene_load_default_tex(rtx_ctx *ctx, const char *bunny_obj_path, unsigned mask, const unsigned char *water_rgb,
                              int water_w, int water_h, const unsigned char *sky_rgb, int sky_w, int sky_h) {
    if (!ctx || !sky_rgb) return hostFail("rt_scene_load_default_tex: null argument");
    const rt::Scene scene;
    return scene.upload(ctx, mask, bunny_obj_path ? bunny_obj_path : "", water_rgb, water_w, water_h, sky_rgb, sky_w, sky_h);
}

int rt_scene_load_default(rtx_ctx *ctx, const char *resources_dir, const char *envmap_path, unsigned mask) {
    if (!ctx || !resources_dir || !envmap_path) return hostFail("rt_scene_load_default: null argument");
    const std::string res = resources_dir;
    unsigned char *water = nullptr, *sky = nullptr;
    int ww = 0, wh = 0, wn = 0, sw = 0, sh = 0, sn = 0;
    int rc = rt_decode_jpeg((res + "/water.jpg").c_str(), &water, &ww, &wh, &wn);
    if (rc == 0 && wn != 3) rc = hostFail("water texture must have 3 components");  // src/Scene.cpp:237
    if (rc == 0) rc = rt_decode_jpeg(envmap_path, &sky, &sw, &sh, &sn);
    if (rc == 0 && sn != 3) rc = hostFail("environment map must have 3 components");  // src/SceneObjects.cpp:424
    if (rc == 0) rc = rt_scene_load_default_tex(ctx, (res + "/bunny.obj").c_str(), mask, water, ww, wh, sky, sw, sh);
    rt_free(water), rt_free(sky);
    return rc;
}

// bitmap_image::save_image, include/bitmap_image.hpp:435-483: "BM", 14-byte file header, 40-byte
// info header, 24 bpp, rows bottom-up padded to 4 bytes.  size_image is computed with the row
// size masked by 0xFFFC (:458), which is kept so files compare equal byte for byte.
int rt_save_bmp(const unsigned char *bgr, int width, int height, const char *path) {
    if (!bgr || !path || width <= 0 || height <= 0) return hostFail("rt_save_bmp: bad arguments");
    FILE *fp = fopen(path, "wb");
    if (!fp) return hostFail(std::string("bitmap_image::save_image(): Error - Could not open file ") + path + " for writing!");
    unsigned char hdr[54];
    auto put16 = [&](int at, unsigned v) { hdr[at] = v & 0xff, hdr[at + 1] = (v >> 8) & 0xff; };
    auto put32 = [&](int at, unsigned v) { put16(at, v & 0xffff), put16(at + 2, v >> 16); };
    const unsigned row = (unsigned)width * 3;
    const unsigned size_image = ((row + 3) & 0x0000FFFCu) * (unsigned)height;
    put16(0, 19778), put32(2, 54 + size_image), put16(6, 0), put16(8, 0), put32(10, 54);
    put32(14, 40), put32(18, (unsigned)width), put32(22, (unsigned)height), put16(26, 1), put16(28, 24);
    put32(30, 0), put32(34, size_image), put32(38, 0), put32(42, 0), put32(46, 0), put32(50, 0);
    bool ok = fwrite(hdr, 1, sizeof hdr, fp) == sizeof hdr;
    const unsigned padding = (4 - (row % 4)) % 4;
    const unsigned char zeros[4] = {0, 0, 0, 0};
    for (int i = 0; i < height && ok; ++i) {
        ok = fwrite(bgr + (size_t)row * (height - i - 1), 1, row, fp) == row;
        if (ok && padding) ok = fwrite(zeros, 1, padding, fp) == padding;
    }
    ok = (fclose(fp) == 0) && ok;
    return ok ? 0 : hostFail("rt_save_bmp: short write");
}

int rt_main(int argc, char **argv) {
    rt::Settings settings;
    try {
        settings = parseArgs(argc, argv);
    } catch (ParseException &e) {
        e.printMessage();
        return -1;
    }
    if (settings.scene != 1) return 0;  // src/rt.cpp:45-46
    const rt::Scene scene;
    return scene.run(settings);
}

}  // extern "C"
