// radish_pt_amd/csrc/host/scene_load.cpp — Radish's scene text format, OBJ meshes and textures → the flattened
// world-space triangle soup the hot path is fed with (C ABI in include/radish_host.h).
//
// Restates, from the reference's sources read as text:
//   Scene::Scene, loadMaterial, loadModel, loadCamera, addTexture      /root/reference/src/scene.cpp:108-141,256-459
//   Scene::buildDevData's flattening loop                                /root/reference/src/scene.cpp:190-223
//   Resource::loadOBJMesh (vertex / normal / texcoord gather)           /root/reference/src/scene.cpp:29-63
//   Math::buildTransformationMatrix                                      /root/reference/src/mathUtil.cpp:12-25
//   utilityCore::tokenizeString / safeGetline                            /root/reference/src/utilities.cpp:58-96
//   Image::Image (stbi_loadf, linear, 3 channels)                        /root/reference/src/image.cpp:14-33
// Third-party code the reference calls here and that is NOT under /root/reference (parity unpinned): glm (matrix
// products / inverse), stb_image (decoding).  tinyobjloader 2.0.0 is vendored (src/tiny_obj_loader.h); what is restated
// of it: `v`/`vn`/`vt`/`f` records, negative indices, triangles as they are, quads split on the shorter diagonal
// (tiny_obj_loader.h:1432-1530).  Polygons with more than four corners are fan-triangulated here (tinyobj ear-clips).
//
// Defined where the reference has undefined behaviour: a ModelInstance without Translate / Rotate / Scale lines gets
// (0,0,0) / (0,0,0) / (1,1,1) (the reference leaves them uninitialised); a face without normals gets its geometric
// normal (the reference reads attrib.normals[-1]).
//
// Decoders built in: PNG (8/16-bit, grey / RGB / palette / alpha, non-interlaced; inflate by zlib), Radiance .hdr (RGBE,
// flat or new-style RLE), binary PPM/PGM (P6/P5), PFM.  Anything else goes to the caller's decode callback (the Python
// mirror passes one built on Pillow).  LDR samples become float v/255 (stbi_ldr_to_hdr_gamma(1.f), scene.cpp:109).
#include <zlib.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/radish_host.h"

namespace {

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
struct Material {  // src/material.h:276-286
    int32_t type = 0;
    float baseColor[3] = {.9f, .9f, .9f};
    float metallic = 0.f, roughness = 1.f, ior = 1.5f;
    int32_t baseColorMapId = -1, normalMapId = -1, metallicMapId = -1, roughnessMapId = -1;
};
static_assert(sizeof(Material) == 44, "Material layout");

struct Error {
    std::string msg;
};
[[noreturn]] void failf(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error{buf};
}

// ---- text helpers (utilities.cpp:58-96) -----------------------------------------------------------------------------
bool safeGetline(std::istream &is, std::string &t) {
    t.clear();
    std::streambuf *sb = is.rdbuf();
    for (;;) {
        int c = sb->sbumpc();
        switch (c) {
        case '\n': return true;
        case '\r':
            if (sb->sgetc() == '\n') sb->sbumpc();
            return true;
        case EOF:
            if (t.empty()) {
                is.setstate(std::ios::eofbit);
                return false;
            }
            return true;
        default: t += (char)c;
        }
    }
}
std::vector<std::string> tokenize(const std::string &s) {
    std::stringstream ss(s);
    std::vector<std::string> out;
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}
float toFloat(const std::vector<std::string> &t, size_t i, const char *what) {
    if (i >= t.size()) failf("%s: missing value", what);
    char *endp = nullptr;
    float v = strtof(t[i].c_str(), &endp);
    if (endp == t[i].c_str()) failf("%s: '%s' is not a number", what, t[i].c_str());
    return v;
}
std::string dirOf(const std::string &path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}
bool fileExists(const std::string &p) {
    std::ifstream f(p.c_str(), std::ios::binary);
    return f.good();
}
std::vector<uint8_t> readFile(const std::string &p) {
    std::ifstream f(p.c_str(), std::ios::binary);
    if (!f) failf("cannot open %s", p.c_str());
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return d;
}
bool endsWithNoCase(const std::string &s, const char *suffix) {
    size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (tolower((unsigned char)s[s.size() - n + i]) != tolower((unsigned char)suffix[i])) return false;
    return true;
}

// ---- images -----------------------------------------------------------------------------------------------------------
struct Image {
    int w = 0, h = 0;
    std::vector<float> rgb;  // row 0 first
};

uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

Image decodePNG(const std::vector<uint8_t> &d, const std::string &name) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) failf("%s: not a PNG", name.c_str());
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    size_t p = 8;
    while (p + 12 <= d.size()) {
        uint32_t len = be32(&d[p]);
        const uint8_t *type = &d[p + 4];
        if (p + 12 + (size_t)len > d.size()) failf("%s: truncated PNG chunk", name.c_str());
        const uint8_t *body = &d[p + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) failf("%s: bad IHDR", name.c_str());
            W = be32(body);
            H = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    if (W == 0 || H == 0 || W > 65535 || H > 65535) failf("%s: bad PNG size", name.c_str());
    if ((uint64_t)W * H > (1ull << 28)) failf("%s: PNG of %u x %u pixels is larger than this loader accepts (2^28)", name.c_str(), W, H);
    if (interlace) failf("%s: interlaced PNG is not supported", name.c_str());
    // the PNG specification's table of bit depths per colour type: 0 (grey) 1,2,4,8,16; 3 (palette) 1,2,4,8; 2, 4, 6: 8 or 16.
    // (A depth of 0 used to give a zero stride and an integer division by zero; depths of 3, 5, 6, 7 were decoded silently.)
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) failf("%s: PNG bit depth %d is not one of 1, 2, 4, 8, 16", name.c_str(), depth);
    if (ctype == 3 && depth == 16) failf("%s: a palette PNG cannot have 16-bit indices", name.c_str());
    if (depth != 8 && depth != 16 && !(ctype == 3 || ctype == 0)) failf("%s: PNG bit depth %d unsupported", name.c_str(), depth);
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) failf("%s: PNG colour type %d unsupported", name.c_str(), ctype);
    if (depth < 8 && !(ctype == 0 || ctype == 3)) failf("%s: PNG bit depth %d unsupported for colour type %d", name.c_str(), depth, ctype);
    size_t bpp = (size_t)(channels * depth + 7) / 8;            // bytes per complete pixel (>= 1) for the filters
    size_t stride = ((size_t)W * channels * depth + 7) / 8;      // bytes per scanline
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        failf("%s: PNG inflate failed", name.c_str());
    std::vector<uint8_t> img(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t *in = &raw[(stride + 1) * y];
        uint8_t *out = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        int filter = in[0];
        in++;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, v = in[i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: {
                int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: failf("%s: bad PNG filter %d", name.c_str(), filter);
            }
            out[i] = (uint8_t)v;
        }
    }
    Image im;
    im.w = (int)W;
    im.h = (int)H;
    im.rgb.resize((size_t)W * H * 3);
    auto sample8 = [&](uint32_t x, uint32_t y, int ch) -> int {  // 8-bit value of channel ch (16-bit → high byte, as stb's 8-bit path)
        const uint8_t *row = &img[stride * y];
        if (depth == 8) return row[(size_t)x * channels + ch];
        if (depth == 16) return row[((size_t)x * channels + ch) * 2];
        size_t bit = (size_t)x * depth;  // 1/2/4-bit grey or palette index
        int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
    };
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            int r, g, b;
            if (ctype == 3) {
                size_t i = (size_t)sample8(x, y, 0) * 3;
                if (i + 2 >= palette.size()) failf("%s: palette index out of range", name.c_str());
                r = palette[i], g = palette[i + 1], b = palette[i + 2];
            } else if (ctype == 0 || ctype == 4) {
                r = g = b = sample8(x, y, 0);
            } else {
                r = sample8(x, y, 0), g = sample8(x, y, 1), b = sample8(x, y, 2);
            }
            float *o = &im.rgb[((size_t)y * W + x) * 3];
            o[0] = r / 255.0f;
            o[1] = g / 255.0f;
            o[2] = b / 255.0f;
        }
    return im;
}

Image decodeHDR(const std::vector<uint8_t> &d, const std::string &name) {
    size_t p = 0;
    auto line = [&]() {
        std::string s;
        while (p < d.size() && d[p] != '\n') s += (char)d[p++];
        p++;
        return s;
    };
    std::string first = line();
    if (first != "#?RADIANCE" && first != "#?RGBE") failf("%s: not a Radiance HDR file", name.c_str());
    bool fmt = false;
    for (;;) {
        if (p >= d.size()) failf("%s: truncated HDR header", name.c_str());
        std::string s = line();
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) failf("%s: unsupported HDR format", name.c_str());
    std::string res = line();
    int H = 0, W = 0;
    if (sscanf(res.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) failf("%s: unsupported HDR orientation '%s'", name.c_str(), res.c_str());
    Image im;
    im.w = W;
    im.h = H;
    im.rgb.resize((size_t)W * H * 3);
    std::vector<uint8_t> scan((size_t)W * 4);
    auto toFloat3 = [](const uint8_t *rgbe, float *o) {  // stbi__hdr_convert
        if (rgbe[3] != 0) {
            float f1 = ldexpf(1.0f, rgbe[3] - (128 + 8));
            o[0] = rgbe[0] * f1;
            o[1] = rgbe[1] * f1;
            o[2] = rgbe[2] * f1;
        } else {
            o[0] = o[1] = o[2] = 0.f;
        }
    };
    for (int y = 0; y < H; y++) {
        if (p + 4 > d.size()) failf("%s: truncated HDR data", name.c_str());
        bool rle = W >= 8 && W < 32768 && d[p] == 2 && d[p + 1] == 2 && !(d[p + 2] & 0x80) && ((d[p + 2] << 8) | d[p + 3]) == W;
        if (!rle) {
            if (p + (size_t)W * 4 > d.size()) failf("%s: truncated HDR data", name.c_str());
            memcpy(scan.data(), &d[p], (size_t)W * 4);
            p += (size_t)W * 4;
        } else {
            p += 4;
            for (int k = 0; k < 4; k++) {
                int x = 0;
                while (x < W) {
                    if (p >= d.size()) failf("%s: truncated HDR data", name.c_str());
                    int count = d[p++];
                    if (count > 128) {
                        count -= 128;
                        if (p >= d.size() || x + count > W) failf("%s: corrupt HDR run", name.c_str());
                        uint8_t v = d[p++];
                        for (int i = 0; i < count; i++) scan[(size_t)(x++) * 4 + k] = v;
                    } else {
                        if (count == 0 || p + (size_t)count > d.size() || x + count > W) failf("%s: corrupt HDR run", name.c_str());
                        for (int i = 0; i < count; i++) scan[(size_t)(x++) * 4 + k] = d[p++];
                    }
                }
            }
        }
        for (int x = 0; x < W; x++) toFloat3(&scan[(size_t)x * 4], &im.rgb[((size_t)y * W + x) * 3]);
    }
    return im;
}

Image decodePNM(const std::vector<uint8_t> &d, const std::string &name) {
    size_t p = 0;
    auto token = [&]() {
        for (;;) {
            while (p < d.size() && isspace(d[p])) p++;
            if (p < d.size() && d[p] == '#') {
                while (p < d.size() && d[p] != '\n') p++;
                continue;
            }
            break;
        }
        std::string s;
        while (p < d.size() && !isspace(d[p])) s += (char)d[p++];
        return s;
    };
    std::string magic = token();
    bool pfm = magic == "PF" || magic == "Pf";
    if (magic != "P6" && magic != "P5" && !pfm) failf("%s: unsupported PNM magic '%s'", name.c_str(), magic.c_str());
    int W = atoi(token().c_str()), H = atoi(token().c_str());
    std::string third = token();
    p++;  // the single whitespace after the header
    if (W <= 0 || H <= 0) failf("%s: bad PNM size", name.c_str());
    int ch = (magic == "P6" || magic == "PF") ? 3 : 1;
    Image im;
    im.w = W;
    im.h = H;
    im.rgb.resize((size_t)W * H * 3);
    if (pfm) {
        float scale = strtof(third.c_str(), nullptr);
        bool little = scale < 0.f;
        if (p + (size_t)W * H * ch * 4 > d.size()) failf("%s: truncated PFM", name.c_str());
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                for (int k = 0; k < 3; k++) {
                    const uint8_t *s = &d[p + (((size_t)(H - 1 - y) * W + x) * ch + (ch == 3 ? k : 0)) * 4];  // PFM rows run bottom to top
                    uint8_t b[4] = {s[0], s[1], s[2], s[3]};
                    if (!little) { b[0] = s[3]; b[1] = s[2]; b[2] = s[1]; b[3] = s[0]; }
                    float v;
                    memcpy(&v, b, 4);
                    im.rgb[((size_t)y * W + x) * 3 + k] = v;
                }
        return im;
    }
    int maxv = atoi(third.c_str());
    if (maxv <= 0 || maxv > 65535) failf("%s: bad PNM maxval", name.c_str());
    int bytes = maxv > 255 ? 2 : 1;
    if (p + (size_t)W * H * ch * bytes > d.size()) failf("%s: truncated PNM", name.c_str());
    for (size_t i = 0; i < (size_t)W * H; i++)
        for (int k = 0; k < 3; k++) {
            const uint8_t *s = &d[p + (i * ch + (ch == 3 ? k : 0)) * bytes];
            int v = bytes == 2 ? ((s[0] << 8) | s[1]) : s[0];
            int v8 = maxv == 255 ? v : (int)((long long)v * 255 / maxv);
            im.rgb[i * 3 + k] = v8 / 255.0f;
        }
    return im;
}

Image loadImage(const std::string &path, bool flipY, rdh_texture_decode_fn decode, void *user) {
    Image im;
    bool native = endsWithNoCase(path, ".png") || endsWithNoCase(path, ".hdr") || endsWithNoCase(path, ".ppm") ||
                  endsWithNoCase(path, ".pgm") || endsWithNoCase(path, ".pfm");
    if (native) {
        std::vector<uint8_t> d = readFile(path);
        if (endsWithNoCase(path, ".png")) im = decodePNG(d, path);
        else if (endsWithNoCase(path, ".hdr")) im = decodeHDR(d, path);
        else im = decodePNM(d, path);
        if (flipY)  // stbi_set_flip_vertically_on_load(true) for everything but the environment map (scene.cpp:110,133-135)
            for (int y = 0; y < im.h / 2; y++)
                for (int x = 0; x < im.w * 3; x++) std::swap(im.rgb[(size_t)y * im.w * 3 + x], im.rgb[(size_t)(im.h - 1 - y) * im.w * 3 + x]);
        return im;
    }
    if (!decode) failf("no decoder for %s (built in: .png .hdr .ppm .pgm .pfm; pass a decode callback for other formats)", path.c_str());
    float *rgb = nullptr;
    int32_t w = 0, h = 0;
    if (decode(path.c_str(), flipY ? 1 : 0, &rgb, &w, &h, user) != 0 || !rgb || w <= 0 || h <= 0)
        failf("Failed to load image %s", path.c_str());  // image.cpp:23-25
    im.w = w;
    im.h = h;
    im.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    free(rgb);
    return im;
}

// ---- OBJ (scene.cpp:29-63 over tinyobjloader) ---------------------------------------------------------------------------
struct Mesh {
    std::vector<V3> vertices, normals;
    std::vector<V2> texcoords;
};
Mesh loadOBJ(const std::string &path) {
    std::ifstream f(path.c_str());
    if (!f) failf("cannot open %s", path.c_str());
    std::vector<V3> v, vn;
    std::vector<V2> vt;
    struct Idx {
        int v, vt, vn;
    };
    std::vector<Idx> corners;  // triangulated
    std::string line;
    auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
    while (safeGetline(f, line) || !line.empty()) {
        std::vector<std::string> t = tokenize(line);
        if (t.empty()) {
            if (!f.good()) break;
            continue;
        }
        if (t[0] == "v" && t.size() >= 4) {
            v.push_back({strtof(t[1].c_str(), nullptr), strtof(t[2].c_str(), nullptr), strtof(t[3].c_str(), nullptr)});
        } else if (t[0] == "vn" && t.size() >= 4) {
            vn.push_back({strtof(t[1].c_str(), nullptr), strtof(t[2].c_str(), nullptr), strtof(t[3].c_str(), nullptr)});
        } else if (t[0] == "vt" && t.size() >= 3) {
            vt.push_back({strtof(t[1].c_str(), nullptr), strtof(t[2].c_str(), nullptr)});
        } else if (t[0] == "f") {
            std::vector<Idx> face;
            for (size_t i = 1; i < t.size(); i++) {
                int a = 0, b = 0, c = 0;
                const char *s = t[i].c_str();
                a = atoi(s);
                const char *s1 = strchr(s, '/');
                if (s1) {
                    if (s1[1] != '/') b = atoi(s1 + 1);
                    const char *s2 = strchr(s1 + 1, '/');
                    if (s2) c = atoi(s2 + 1);
                }
                Idx id{fix(a, v.size()), fix(b, vt.size()), fix(c, vn.size())};
                if (id.v < 0 || id.v >= (int)v.size()) failf("%s: face references vertex %d of %zu", path.c_str(), a, v.size());
                if (id.vt >= (int)vt.size() || id.vn >= (int)vn.size()) failf("%s: face index out of range", path.c_str());
                face.push_back(id);
            }
            if (face.size() < 3) continue;  // "Degenerated face found"
            if (face.size() == 3) {
                corners.insert(corners.end(), face.begin(), face.end());
            } else if (face.size() == 4) {  // split on the shorter diagonal (tiny_obj_loader.h:1484-1530)
                V3 p0 = v[face[0].v], p1 = v[face[1].v], p2 = v[face[2].v], p3 = v[face[3].v];
                float e02x = p2.x - p0.x, e02y = p2.y - p0.y, e02z = p2.z - p0.z;
                float e13x = p3.x - p1.x, e13y = p3.y - p1.y, e13z = p3.z - p1.z;
                float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z, sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                const int a[6] = {0, 1, 2, 0, 2, 3}, b[6] = {0, 1, 3, 1, 2, 3};
                for (int k = 0; k < 6; k++) corners.push_back(face[sqr02 < sqr13 ? a[k] : b[k]]);
            } else {
                for (size_t k = 1; k + 1 < face.size(); k++) {
                    corners.push_back(face[0]);
                    corners.push_back(face[k]);
                    corners.push_back(face[k + 1]);
                }
            }
        }
        if (!f.good() && line.empty()) break;
    }
    Mesh m;
    const bool hasTexcoord = !vt.empty();
    for (size_t i = 0; i < corners.size(); i += 3) {
        V3 p[3] = {v[corners[i].v], v[corners[i + 1].v], v[corners[i + 2].v]};
        V3 e1{p[1].x - p[0].x, p[1].y - p[0].y, p[1].z - p[0].z}, e2{p[2].x - p[0].x, p[2].y - p[0].y, p[2].z - p[0].z};
        V3 gn{e1.y * e2.z - e2.y * e1.z, e1.z * e2.x - e2.z * e1.x, e1.x * e2.y - e2.x * e1.y};
        float len = sqrtf(gn.x * gn.x + gn.y * gn.y + gn.z * gn.z);
        if (len > 0.f) gn = {gn.x / len, gn.y / len, gn.z / len};
        for (int k = 0; k < 3; k++) {
            const Idx &c = corners[i + k];
            m.vertices.push_back(p[k]);
            m.normals.push_back(c.vn >= 0 ? vn[c.vn] : gn);
            m.texcoords.push_back((hasTexcoord && c.vt >= 0) ? vt[c.vt] : V2{0.f, 0.f});
        }
    }
    return m;
}

// ---- transforms (mathUtil.cpp:12-25 over glm) ---------------------------------------------------------------------------
struct M4 {
    float m[4][4];  // m[col][row], like glm
};
M4 identity() {
    M4 r{};
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.f;
    return r;
}
M4 mul(const M4 &a, const M4 &b) {  // glm operator*: result column j = a * b[j]
    M4 r{};
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) r.m[j][i] = a.m[0][i] * b.m[j][0] + a.m[1][i] * b.m[j][1] + a.m[2][i] * b.m[j][2] + a.m[3][i] * b.m[j][3];
    return r;
}
M4 rotateAxis(float angle, int axis) {  // glm::rotate(mat4(1), angle, unit axis)
    float c = cosf(angle), s = sinf(angle);
    M4 r = identity();
    int a = (axis + 1) % 3, b = (axis + 2) % 3;
    r.m[a][a] = c;
    r.m[a][b] = s;
    r.m[b][a] = -s;
    r.m[b][b] = c;
    return r;
}
bool inverse4(const M4 &in, M4 &out) {  // cofactor expansion (glm::inverse's formula)
    const float *m = &in.m[0][0];
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.f) return false;
    float od = 1.f / det;
    for (int i = 0; i < 16; i++) (&out.m[0][0])[i] = inv[i] * od;
    return true;
}

struct Parsed {
    std::vector<float> vertices, normals, texcoords;
    std::vector<int32_t> materialIds;
    std::vector<Material> materials;
    std::vector<Image> images;
    std::vector<rdh_host_texture> textures;
    rdh_parsed_scene pub{};
};

struct Loader {
    std::ifstream fp;
    std::string baseDir;
    rdh_texture_decode_fn decode = nullptr;
    void *user = nullptr;
    Parsed *out = nullptr;
    std::map<std::string, int> materialMap, textureMap;
    std::map<std::string, Mesh> meshPool;
    bool flipTextures = true;

    std::string resolve(const std::string &p) const {  // the reference opens paths relative to its working directory;
        if (fileExists(p)) return p;                   // also try relative to the scene file
        std::string q = baseDir + p;
        return fileExists(q) ? q : p;
    }
    int addTexture(const std::string &name) {  // scene.cpp:395-406 + Resource::loadTexture's pool (:79-87)
        auto it = textureMap.find(name);
        if (it != textureMap.end()) return it->second;
        out->images.push_back(loadImage(resolve(name), flipTextures, decode, user));
        int id = (int)out->images.size() - 1;
        textureMap[name] = id;
        return id;
    }
    void loadMaterial(const std::string &id) {  // scene.cpp:408-459
        Material m;
        for (int i = 0; i < 6; i++) {
            std::string line;
            safeGetline(fp, line);
            std::vector<std::string> t = tokenize(line);
            if (t.size() < 2) failf("Material %s: expected 6 property lines, line %d is '%s'", id.c_str(), i + 1, line.c_str());
            if (t[0] == "Type") {
                static const std::map<std::string, int> types = {{"Lambertian", 0}, {"MetallicWorkflow", 1}, {"Dielectric", 2}, {"Light", 4}};
                auto f = types.find(t[1]);
                m.type = f == types.end() ? 0 : f->second;  // std::map::operator[] default-inserts 0 = Lambertian (scene.cpp:413)
            } else if (t[0] == "BaseColor") {
                if (t.size() > 2) {
                    for (int k = 0; k < 3; k++) m.baseColor[k] = toFloat(t, 1 + k, "BaseColor");
                } else if (t[1] == "Procedural") {
                    m.baseColorMapId = -2;
                } else {
                    m.baseColorMapId = addTexture(t[1]);
                }
            } else if (t[0] == "Metallic") {
                if (isdigit((unsigned char)t[1][t[1].length() - 1])) m.metallic = toFloat(t, 1, "Metallic");
                else m.metallicMapId = addTexture(t[1]);
            } else if (t[0] == "Roughness") {
                if (isdigit((unsigned char)t[1][t[1].length() - 1])) m.roughness = toFloat(t, 1, "Roughness");
                else m.roughnessMapId = addTexture(t[1]);
            } else if (t[0] == "Ior") {
                m.ior = toFloat(t, 1, "Ior");
            } else if (t[0] == "NormalMap") {
                if (t[1] != "Null") m.normalMapId = addTexture(t[1]);
            }
        }
        materialMap[id] = (int)out->materials.size();
        out->materials.push_back(m);
    }
    void loadModel() {  // scene.cpp:256-317 + buildDevData's flattening (:190-223)
        std::string line;
        safeGetline(fp, line);
        std::string filename = line;
        while (!filename.empty() && isspace((unsigned char)filename.back())) filename.pop_back();
        const Mesh *mesh = nullptr;
        if (filename.find(".obj") != std::string::npos) {
            auto it = meshPool.find(filename);
            if (it == meshPool.end()) {
                std::string path = resolve(filename);
                if (fileExists(path)) it = meshPool.emplace(filename, loadOBJ(path)).first;
            }
            if (it != meshPool.end()) mesh = &it->second;
        }
        if (!mesh) {  // "[Fail to load, skipped]"
            while (!line.empty() && fp.good()) safeGetline(fp, line);
            return;
        }
        int materialId = -1;
        safeGetline(fp, line);
        if (!line.empty()) {
            std::vector<std::string> t = tokenize(line);
            if (t.size() < 2) failf("Object: material line '%s' needs two tokens", line.c_str());
            if (t[1] == "Null") {
                out->materials.push_back(Material());
                materialId = (int)out->materials.size() - 1;
            } else {
                auto f = materialMap.find(t[1]);
                if (f == materialMap.end()) failf("Material %s not found!", t[1].c_str());
                materialId = f->second;
            }
        }
        if (materialId < 0) failf("Object %s has no material line", filename.c_str());
        float tr[3] = {0.f, 0.f, 0.f}, rot[3] = {0.f, 0.f, 0.f}, sc[3] = {1.f, 1.f, 1.f};
        safeGetline(fp, line);
        while (!line.empty()) {
            std::vector<std::string> t = tokenize(line);
            if (!t.empty()) {
                float *dst = t[0] == "Translate" ? tr : t[0] == "Rotate" ? rot : t[0] == "Scale" ? sc : nullptr;
                if (dst)
                    for (int k = 0; k < 3; k++) dst[k] = toFloat(t, 1 + k, t[0].c_str());
            }
            if (!fp.good()) break;
            safeGetline(fp, line);
        }
        const float PI = 3.1415926535897932384626422832795028841971f;
        M4 T = identity();
        T.m[3][0] = tr[0];
        T.m[3][1] = tr[1];
        T.m[3][2] = tr[2];
        M4 R = mul(mul(rotateAxis(rot[0] * PI / 180.f, 0), rotateAxis(rot[1] * PI / 180.f, 1)), rotateAxis(rot[2] * PI / 180.f, 2));
        M4 S = identity();
        S.m[0][0] = sc[0];
        S.m[1][1] = sc[1];
        S.m[2][2] = sc[2];
        M4 X = mul(mul(T, R), S), Xi;
        if (!inverse4(X, Xi)) failf("Object %s: singular transform", filename.c_str());
        // normalMatrix = transpose(mat3(inverse)): n' = normalize(N * n), N[col][row] = Xi[row][col]
        for (size_t i = 0; i < mesh->vertices.size(); i++) {
            V3 p = mesh->vertices[i], n = mesh->normals[i];
            float w[3], nn[3];
            for (int r = 0; r < 3; r++) {
                w[r] = X.m[0][r] * p.x + X.m[1][r] * p.y + X.m[2][r] * p.z + X.m[3][r] * 1.0f;
                nn[r] = Xi.m[r][0] * n.x + Xi.m[r][1] * n.y + Xi.m[r][2] * n.z;
            }
            float inv = 1.f / sqrtf((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]);
            out->vertices.insert(out->vertices.end(), {w[0], w[1], w[2]});
            out->normals.insert(out->normals.end(), {nn[0] * inv, nn[1] * inv, nn[2] * inv});
            out->texcoords.insert(out->texcoords.end(), {mesh->texcoords[i].x, mesh->texcoords[i].y});
            if (i % 3 == 0) out->materialIds.push_back(materialId);
        }
    }
    void loadCamera() {  // scene.cpp:319-393
        struct Cam {
            int32_t resx, resy;
            float position[3], rotation[3], view[3], up[3], right[3];
            float fovx, fovy, pixLenX, pixLenY;
            float rotInv[9];
            float viewProj[16];
            float lensRadius, focalDist, tanFovY;
        } c;
        static_assert(sizeof(Cam) == 196, "Camera layout");
        memset(&c, 0, sizeof(c));
        float fovy = 0.f;
        for (int i = 0; i < 8; i++) {
            std::string line;
            safeGetline(fp, line);
            std::vector<std::string> t = tokenize(line);
            if (t.size() < 2) failf("Camera: expected 8 property lines, line %d is '%s'", i + 1, line.c_str());
            if (t[0] == "Resolution") {
                c.resx = atoi(t[1].c_str());
                c.resy = t.size() > 2 ? atoi(t[2].c_str()) : 0;
            } else if (t[0] == "FovY") fovy = toFloat(t, 1, "FovY");
            else if (t[0] == "LensRadius") c.lensRadius = toFloat(t, 1, "LensRadius");
            else if (t[0] == "FocalDist") c.focalDist = toFloat(t, 1, "FocalDist");
            else if (t[0] == "ApertureMask") {
                if (t[1] != "Null") out->pub.apertureMaskTexId = addTexture(t[1]);
            } else if (t[0] == "Sample") out->pub.iterations = atoi(t[1].c_str());
            else if (t[0] == "Depth") out->pub.traceDepth = atoi(t[1].c_str());
            else if (t[0] == "File") snprintf(out->pub.imageName, sizeof(out->pub.imageName), "%s", t[1].c_str());
        }
        std::string line;
        safeGetline(fp, line);
        while (!line.empty()) {
            std::vector<std::string> t = tokenize(line);
            if (!t.empty()) {
                float *dst = t[0] == "Eye" ? c.position : t[0] == "Rotation" ? c.rotation : t[0] == "Up" ? c.up : nullptr;
                if (dst)
                    for (int k = 0; k < 3; k++) dst[k] = toFloat(t, 1 + k, t[0].c_str());
            }
            if (!fp.good()) break;
            safeGetline(fp, line);
        }
        if (c.resx <= 0 || c.resy <= 0) failf("Camera: bad Resolution %d x %d", c.resx, c.resy);
        c.fovy = fovy;
        memcpy(out->pub.camera, &c, sizeof(c));
        rdh_camera_update(out->pub.camera);  // fov.x, tanFovY (scene.cpp:378-383) + Camera::update (sceneStructs.h:93-107)
        out->pub.hasCamera = 1;
    }
    void run(const std::string &file) {  // Scene::Scene (scene.cpp:108-141)
        fp.open(file.c_str());
        if (!fp.is_open()) failf("Error reading from file %s - aborting!", file.c_str());
        baseDir = dirOf(file);
        while (fp.good()) {
            std::string line;
            safeGetline(fp, line);
            if (line.empty()) continue;
            std::vector<std::string> t = tokenize(line);
            if (t.empty()) continue;
            if (t[0] == "Material") {
                if (t.size() < 2) failf("Material block without a name");
                loadMaterial(t[1]);
            } else if (t[0] == "Object") {
                loadModel();
            } else if (t[0] == "Camera") {
                loadCamera();
            } else if (t[0] == "EnvMap") {
                if (t.size() > 1 && t[1] != "Null") {
                    flipTextures = false;
                    out->pub.envMapTexId = addTexture(t[1]);
                    flipTextures = true;
                }
            }
        }
    }
};

}  // namespace

extern "C" int32_t rdh_scene_parse(const char *sceneFile, rdh_texture_decode_fn decode, void *user, rdh_parsed_scene **outScene,
                                   char *err, int32_t errLen) {
    if (err && errLen > 0) err[0] = 0;
    if (!sceneFile || !outScene) return RDH_HOST_ERR_ARGS;
    *outScene = nullptr;
    Parsed *p = new Parsed;
    p->pub.envMapTexId = -1;
    p->pub.apertureMaskTexId = -1;
    try {
        Loader L;
        L.decode = decode;
        L.user = user;
        L.out = p;
        L.run(sceneFile);
        if (p->materialIds.empty()) failf("[No mesh data loaded, quit]");  // scene.cpp:225-228
    } catch (const Error &e) {
        if (err && errLen > 0) snprintf(err, (size_t)errLen, "%s", e.msg.c_str());
        delete p;
        return RDH_HOST_ERR_SCENE;
    } catch (const std::exception &e) {
        if (err && errLen > 0) snprintf(err, (size_t)errLen, "%s", e.what());
        delete p;
        return RDH_HOST_ERR_SCENE;
    }
    for (const Image &im : p->images) p->textures.push_back(rdh_host_texture{im.w, im.h, im.rgb.data()});
    rdh_parsed_scene &s = p->pub;
    s.numPrims = (int32_t)p->materialIds.size();
    s.vertices = p->vertices.data();
    s.normals = p->normals.data();
    s.texcoords = p->texcoords.data();
    s.materialIds = p->materialIds.data();
    s.numMaterials = (int32_t)p->materials.size();
    s.materials = p->materials.data();
    s.numTextures = (int32_t)p->textures.size();
    s.textures = p->textures.data();
    s.opaque = p;
    *outScene = &p->pub;
    return 0;
}

extern "C" void rdh_scene_parse_free(rdh_parsed_scene *s) {
    if (s) delete static_cast<Parsed *>(s->opaque);
}
