// ply.cpp -- scene ingest: the input side of `render(ply, camera) -> image`.
// Restates what read_gs_ply does (app/gaussians.cpp:75-171) without the vendored happly parser:
// properties are looked up BY NAME in the `vertex` element (x y z, f_dc_0..2, f_rest_0..44, opacity,
// scale_0..2, rot_0..3; anything else, e.g. nx ny nz, is ignored), then
//   opacity = sigmoid(raw)            (gaussians.cpp:15-19,140)
//   scale   = exp(raw)                (gaussians.cpp:21-25,150)
//   rotq    = raw / |raw|, (r,x,y,z)  (gaussians.cpp:27-35,154-168)
//   feature[j*48 + k*3 + c]: f_dc_c -> k = 0; f_rest_i -> c = i / 15, k = i % 15 + 1   (gaussians.cpp:106-135)
// The reference copies 62 whole columns through std::vector<float>; here the file is read once and
// de-interleaved in a single multithreaded pass over the vertex records.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../common.hpp"

namespace
{

struct Prop {
    std::string name;
    int         type; // 0 i8, 1 u8, 2 i16, 3 u16, 4 i32, 5 u32, 6 f32, 7 f64
    size_t      offset;
};

int type_from_name(const std::string& t)
{
    if (t == "char" || t == "int8") return 0;
    if (t == "uchar" || t == "uint8") return 1;
    if (t == "short" || t == "int16") return 2;
    if (t == "ushort" || t == "uint16") return 3;
    if (t == "int" || t == "int32") return 4;
    if (t == "uint" || t == "uint32") return 5;
    if (t == "float" || t == "float32") return 6;
    if (t == "double" || t == "float64") return 7;
    return -1;
}
const size_t kTypeSize[8] = { 1, 1, 2, 2, 4, 4, 4, 8 };

inline float load_as_float(const unsigned char* p, int type)
{
    switch (type) {
    case 0: return (float)*reinterpret_cast<const signed char*>(p);
    case 1: return (float)*p;
    case 2: { int16_t v; memcpy(&v, p, 2); return (float)v; }
    case 3: { uint16_t v; memcpy(&v, p, 2); return (float)v; }
    case 4: { int32_t v; memcpy(&v, p, 4); return (float)v; }
    case 5: { uint32_t v; memcpy(&v, p, 4); return (float)v; }
    case 6: { float v; memcpy(&v, p, 4); return v; }
    default: { double v; memcpy(&v, p, 8); return (float)v; }
    }
}

template <typename F>
void parallel_for(int64_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (n < 65536) nt = 1;
    if (nt > 32) nt = 32;
    std::vector<std::thread> th;
    int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        int64_t a = (int64_t)t * chunk, b = std::min<int64_t>(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=] { f(a, b); });
    }
    for (auto& x : th) x.join();
}

lcgs_status fail(lcgs_status s, const std::string& msg)
{
    lcgs::set_last_error(msg);
    return s;
}

// Parses the header (fp is left at the first payload byte) and looks the 59 wanted columns up by name.
lcgs_status parse_header(FILE* fp, std::vector<Prop>& props, int64_t& N, size_t& stride, bool& binary,
                         std::vector<int>& want)
{
    props.clear();
    want.clear();
    stride = 0;
    N      = -1;
    binary = false;
    std::string line;
    bool        ascii = false, in_vertex = false, seen_vertex = false, header_ok = false;
    bool        first = true;
    for (;;) {
        char buf[1024];
        if (!fgets(buf, sizeof(buf), fp)) break;
        line = buf;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (first) {
            first = false;
            if (line != "ply") {
                return fail(LCGS_ERR_FORMAT, "not a PLY file (missing magic)");
            }
            continue;
        }
        std::istringstream ss(line);
        std::string        tok;
        ss >> tok;
        if (tok == "format") {
            std::string f;
            ss >> f;
            if (f == "binary_little_endian") binary = true;
            else if (f == "ascii") ascii = true;
            else {
                return fail(LCGS_ERR_FORMAT, "unsupported PLY format: " + f);
            }
        } else if (tok == "element") {
            std::string name;
            int64_t     cnt;
            ss >> name >> cnt;
            if (name == "vertex") {
                if (seen_vertex) {
                        return fail(LCGS_ERR_FORMAT, "duplicate vertex element");
                }
                in_vertex = seen_vertex = true;
                N                       = cnt;
            } else {
                if (!seen_vertex && cnt > 0) {
                        return fail(LCGS_ERR_FORMAT, "elements before `vertex` are not supported");
                }
                in_vertex = false;
            }
        } else if (tok == "property") {
            if (!in_vertex) continue;
            std::string t, name;
            ss >> t;
            if (t == "list") {
                return fail(LCGS_ERR_FORMAT, "list property in vertex element");
            }
            ss >> name;
            int ty = type_from_name(t);
            if (ty < 0) {
                return fail(LCGS_ERR_FORMAT, "unknown property type: " + t);
            }
            props.push_back({ name, ty, stride });
            stride += kTypeSize[ty];
        } else if (tok == "end_header") {
            header_ok = true;
            break;
        }
    }
    if (!header_ok || !seen_vertex || N < 0 || (!binary && !ascii)) {
        return fail(LCGS_ERR_FORMAT, "No vertex element in the ply file"); // gaussians.cpp:80-82
    }
    if (N >= (1 << 30)) {
        return fail(LCGS_ERR_FORMAT, "too many vertices");
    }

    // ---- property lookup by name (happly throws on a missing one, app/happly.h:974)
    auto find = [&](const std::string& name) -> int {
        for (size_t i = 0; i < props.size(); ++i)
            if (props[i].name == name) return (int)i;
        return -1;
    };
    // 59 columns in the order pos(3) dc(3) rest(45) opacity scale(3) rot(4)
    std::vector<std::string> names = { "x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2" };
    for (int i = 0; i < 45; ++i) names.push_back("f_rest_" + std::to_string(i));
    names.push_back("opacity");
    for (int i = 0; i < 3; ++i) names.push_back("scale_" + std::to_string(i));
    for (int i = 0; i < 4; ++i) names.push_back("rot_" + std::to_string(i));
    for (auto& n : names) {
        int k = find(n);
        if (k < 0) {
            return fail(LCGS_ERR_FORMAT, "PLY vertex element has no property `" + n + "`");
        }
        want.push_back(k);
    }

    return LCGS_OK;
}

} // namespace

extern "C" {

void lcgs_scene_host_free(lcgs_scene_host* s)
{
    if (!s) return;
    free(s->pos);
    free(s->feature);
    free(s->opacity);
    free(s->scale);
    free(s->rotq);
    memset(s, 0, sizeof(*s));
}

lcgs_status lcgs_ply_read(const char* path, lcgs_scene_host* out)
{
    if (!path || !out) return fail(LCGS_ERR_INVALID_ARG, "lcgs_ply_read: NULL argument");
    memset(out, 0, sizeof(*out));
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(LCGS_ERR_IO, std::string("cannot open ") + path);

    std::vector<Prop> props;
    std::vector<int>  want;
    int64_t           N = -1;
    size_t            stride = 0;
    bool              binary = false;
    {
        lcgs_status hs = parse_header(fp, props, N, stride, binary, want);
        if (hs != LCGS_OK) {
            fclose(fp);
            return hs;
        }
    }

    // ---- payload
    const size_t           np = props.size();
    std::vector<unsigned char> raw;
    std::vector<float>     asc;
    if (binary) {
        raw.resize((size_t)N * stride);
        size_t got = fread(raw.data(), 1, raw.size(), fp);
        if (got != raw.size()) {
            fclose(fp);
            return fail(LCGS_ERR_FORMAT, "PLY payload is truncated");
        }
    } else {
        asc.resize((size_t)N * np);
        for (size_t i = 0; i < asc.size(); ++i) {
            double v;
            if (fscanf(fp, "%lf", &v) != 1) {
                fclose(fp);
                return fail(LCGS_ERR_FORMAT, "PLY ascii payload is truncated");
            }
            asc[i] = (float)v;
        }
    }
    fclose(fp);

    out->num_gaussians = (int)N;
    out->sh_degree     = 3; // app/gaussians.h:17
    size_t n1          = (size_t)std::max<int64_t>(N, 1);
    out->pos           = (float*)malloc(n1 * 3 * sizeof(float));
    out->feature       = (float*)malloc(n1 * 48 * sizeof(float));
    out->opacity       = (float*)malloc(n1 * sizeof(float));
    out->scale         = (float*)malloc(n1 * 3 * sizeof(float));
    out->rotq          = (float*)malloc(n1 * 4 * sizeof(float));
    if (!out->pos || !out->feature || !out->opacity || !out->scale || !out->rotq) {
        lcgs_scene_host_free(out);
        return fail(LCGS_ERR_OUT_OF_MEMORY, "host allocation failed");
    }

    parallel_for(N, [&](int64_t a, int64_t b) {
        for (int64_t j = a; j < b; ++j) {
            auto col = [&](int w) -> float {
                const int k = want[w];
                if (binary) return load_as_float(raw.data() + (size_t)j * stride + props[k].offset, props[k].type);
                return asc[(size_t)j * np + k];
            };
            for (int c = 0; c < 3; ++c) out->pos[3 * j + c] = col(c);
            float* f = out->feature + (size_t)j * 48;
            for (int c = 0; c < 3; ++c) f[0 * 3 + c] = col(3 + c);
            for (int i = 0; i < 45; ++i) {
                const int channel = i / 15, offset = i % 15 + 1;
                f[offset * 3 + channel] = col(6 + i);
            }
            out->opacity[j] = 1.0f / (1.0f + expf(-col(51)));
            for (int c = 0; c < 3; ++c) out->scale[3 * j + c] = expf(col(52 + c));
            float r = col(55), x = col(56), y = col(57), z = col(58);
            float norm = sqrtf(x * x + y * y + z * z + r * r);
            out->rotq[4 * j + 0] = r / norm;
            out->rotq[4 * j + 1] = x / norm;
            out->rotq[4 * j + 2] = y / norm;
            out->rotq[4 * j + 3] = z / norm;
        }
    });
    return LCGS_OK;
}

lcgs_status lcgs_ply_write_raw(const char* path, int num_gaussians, const float* pos, const float* f_dc,
                               const float* f_rest, const float* opacity_logit, const float* log_scale,
                               const float* rot)
{
    if (!path || num_gaussians < 0) return fail(LCGS_ERR_INVALID_ARG, "lcgs_ply_write_raw: bad argument");
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(LCGS_ERR_IO, std::string("cannot open ") + path + " for writing");
    fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %d\n", num_gaussians);
    const char* head[] = { "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2" };
    for (auto h : head) fprintf(fp, "property float %s\n", h);
    for (int i = 0; i < 45; ++i) fprintf(fp, "property float f_rest_%d\n", i);
    fprintf(fp, "property float opacity\n");
    for (int i = 0; i < 3; ++i) fprintf(fp, "property float scale_%d\n", i);
    for (int i = 0; i < 4; ++i) fprintf(fp, "property float rot_%d\n", i);
    fprintf(fp, "end_header\n");
    std::vector<float> rec(62);
    for (int j = 0; j < num_gaussians; ++j) {
        int k = 0;
        for (int c = 0; c < 3; ++c) rec[k++] = pos[3 * (size_t)j + c];
        for (int c = 0; c < 3; ++c) rec[k++] = 0.0f;
        for (int c = 0; c < 3; ++c) rec[k++] = f_dc[3 * (size_t)j + c];
        for (int i = 0; i < 45; ++i) rec[k++] = f_rest[45 * (size_t)j + i];
        rec[k++] = opacity_logit[j];
        for (int c = 0; c < 3; ++c) rec[k++] = log_scale[3 * (size_t)j + c];
        for (int c = 0; c < 4; ++c) rec[k++] = rot[4 * (size_t)j + c];
        if (fwrite(rec.data(), sizeof(float), 62, fp) != 62) {
            fclose(fp);
            return fail(LCGS_ERR_IO, "short write");
        }
    }
    fclose(fp);
    return LCGS_OK;
}

} // extern "C"

namespace lcgs
{

// Header of a binary-little-endian 3DGS PLY for the device ingest path (lcgs_scene_load_ply): record count, stride,
// payload offset and the byte offsets of the 59 wanted columns.  device_ok is false when the file needs the host
// path (ascii, or a wanted column that is not float32).
lcgs_status ply_probe(const char* path, PlyProbe* out)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(LCGS_ERR_IO, std::string("cannot open ") + path);
    std::vector<Prop> props;
    std::vector<int>  want;
    bool              binary = false;
    size_t            stride = 0;
    lcgs_status       hs     = parse_header(fp, props, out->num_vertices, stride, binary, want);
    if (hs != LCGS_OK) {
        fclose(fp);
        return hs;
    }
    out->stride         = stride;
    out->payload_offset = (size_t)ftell(fp);
    out->device_ok      = binary && stride % 4 == 0;
    for (int w = 0; w < 59; ++w) {
        out->column_offset[w] = (uint32_t)props[want[w]].offset;
        if (props[want[w]].type != 6 || props[want[w]].offset % 4 != 0) out->device_ok = false;
    }
    fseek(fp, 0, SEEK_END);
    const size_t file_size = (size_t)ftell(fp);
    fclose(fp);
    if (binary && file_size < out->payload_offset + (size_t)out->num_vertices * stride)
        return fail(LCGS_ERR_FORMAT, "PLY payload is truncated");
    return LCGS_OK;
}

} // namespace lcgs

