// Sanitizer driver for the host-side PLY reader (tests/test_host_sanitizers.py builds it with the product's
// csrc/host/ply.cpp under -fsanitize=address,undefined).  TEST HELPER, not part of the product library.
// Feeds the reader well-formed files (binary, ascii, extra properties, double columns) and malformed ones (truncated
// payload, truncated header, bad magic, absurd counts, list properties, missing columns, empty file); every call must
// return a status -- never read out of bounds, never leak, never overflow.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../luisacomputegaussiansplatting_amd/csrc/host/ply.cpp"

namespace lcgs
{
static std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
} // namespace lcgs

static std::vector<std::string> wanted()
{
    std::vector<std::string> n = { "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2" };
    for (int i = 0; i < 45; ++i) n.push_back("f_rest_" + std::to_string(i));
    n.push_back("opacity");
    for (int i = 0; i < 3; ++i) n.push_back("scale_" + std::to_string(i));
    for (int i = 0; i < 4; ++i) n.push_back("rot_" + std::to_string(i));
    return n;
}

static void write_file(const char* path, const std::string& header, const std::vector<unsigned char>& payload)
{
    FILE* f = fopen(path, "wb");
    fwrite(header.data(), 1, header.size(), f);
    if (!payload.empty()) fwrite(payload.data(), 1, payload.size(), f);
    fclose(f);
}

static int expect(const char* path, bool ok, int n = -1)
{
    lcgs_scene_host s;
    lcgs_status     st = lcgs_ply_read(path, &s);
    int             bad = 0;
    if (ok != (st == LCGS_OK)) bad = 1;
    if (st == LCGS_OK && n >= 0 && s.num_gaussians != n) bad = 1;
    if (st == LCGS_OK) lcgs_scene_host_free(&s);
    lcgs::PlyProbe pr;
    (void)lcgs::ply_probe(path, &pr);
    if (bad) fprintf(stderr, "unexpected result for %s: status %d (%s)\n", path, (int)st, lcgs::g_err.c_str());
    return bad;
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const auto        names = wanted();
    int               failures = 0;
    auto P = [&](const char* n) { return dir + "/" + n; };
    auto header = [&](const char* fmt, long long count, const std::vector<std::string>& props, const char* type = "float") {
        std::string h = std::string("ply\nformat ") + fmt + " 1.0\nelement vertex " + std::to_string(count) + "\n";
        for (auto& p : props) h += std::string("property ") + type + " " + p + "\n";
        return h + "end_header\n";
    };
    const int N = 37;
    std::vector<unsigned char> payload((size_t)N * names.size() * 4);
    for (size_t i = 0; i < payload.size(); ++i) payload[i] = (unsigned char)(i * 2654435761u >> 13);
    // well-formed
    write_file(P("ok.ply").c_str(), header("binary_little_endian", N, names), payload);
    failures += expect(P("ok.ply").c_str(), true, N);
    write_file(P("empty_scene.ply").c_str(), header("binary_little_endian", 0, names), {});
    failures += expect(P("empty_scene.ply").c_str(), true, 0);
    {
        std::vector<unsigned char> dbl((size_t)N * names.size() * 8, 0);
        write_file(P("double.ply").c_str(), header("binary_little_endian", N, names, "double"), dbl);
        failures += expect(P("double.ply").c_str(), true, N);
    }
    {
        std::string body;
        for (int j = 0; j < 3; ++j) {
            for (size_t k = 0; k < names.size(); ++k) body += std::to_string(0.01 * (double)(j + k)) + " ";
            body += "\n";
        }
        std::vector<unsigned char> b(body.begin(), body.end());
        write_file(P("ascii.ply").c_str(), header("ascii", 3, names), b);
        failures += expect(P("ascii.ply").c_str(), true, 3);
        b.resize(b.size() / 2);
        write_file(P("ascii_trunc.ply").c_str(), header("ascii", 3, names), b);
        failures += expect(P("ascii_trunc.ply").c_str(), false);
    }
    // malformed
    {
        auto t = payload;
        t.resize(t.size() - 5);
        write_file(P("trunc.ply").c_str(), header("binary_little_endian", N, names), t);
        failures += expect(P("trunc.ply").c_str(), false);
    }
    write_file(P("huge.ply").c_str(), header("binary_little_endian", 2000000000LL, names), payload);
    failures += expect(P("huge.ply").c_str(), false);
    write_file(P("negative.ply").c_str(), header("binary_little_endian", -5, names), payload);
    failures += expect(P("negative.ply").c_str(), false);
    {
        auto fewer = names;
        fewer.pop_back();
        write_file(P("missing.ply").c_str(), header("binary_little_endian", N, fewer), payload);
        failures += expect(P("missing.ply").c_str(), false);
    }
    write_file(P("magic.ply").c_str(), "plx\nformat binary_little_endian 1.0\nend_header\n", {});
    failures += expect(P("magic.ply").c_str(), false);
    write_file(P("nohdr.ply").c_str(), "ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\n", {});
    failures += expect(P("nohdr.ply").c_str(), false);
    write_file(P("list.ply").c_str(), "ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty list uchar int idx\nend_header\n", {});
    failures += expect(P("list.ply").c_str(), false);
    write_file(P("bigendian.ply").c_str(), header("binary_big_endian", N, names), payload);
    failures += expect(P("bigendian.ply").c_str(), false);
    write_file(P("zero.ply").c_str(), "", {});
    failures += expect(P("zero.ply").c_str(), false);
    failures += expect(P("does_not_exist.ply").c_str(), false);
    // the writer and its round trip
    {
        std::vector<float> pos(3 * N, 0.5f), dc(3 * N, 0.1f), rest(45 * N, 0.01f), op(N, 0.0f), ls(3 * N, -4.0f), rot(4 * N, 0.5f);
        if (lcgs_ply_write_raw(P("rt.ply").c_str(), N, pos.data(), dc.data(), rest.data(), op.data(), ls.data(), rot.data()) != LCGS_OK) ++failures;
        failures += expect(P("rt.ply").c_str(), true, N);
    }
    printf("failures %d\n", failures);
    return failures ? 1 : 0;
}
