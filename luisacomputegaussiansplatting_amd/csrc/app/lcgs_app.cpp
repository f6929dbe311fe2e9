// lcgs-app -- CLI work-alike of the reference's app/main.cpp on the MI355X library.
//   lcgs-app --ply <path> [--res WxH] [--out dir] [--world colmap|blender] [--exp_N N] [--backend hip]
//            [--path fused|stage|deferred] [--synth kind:count:seed] [--ingest device|host] [--cameras file]
//            [--order file|spatial] [--pose garden|lego] [--gpus N] [--backward [--owner] [--comm-selftest]] [--fit K]
// Same flags (app/main.cpp:52-124; `--key=value` and `--key value`, app/command_parser.hpp:5-79), the same
// hard-coded look-at camera (app/main.cpp:191-207), the same frame loop (:266-308), the same output:
// <out>/<ply stem>_<backend>.png, CHW float -> vertically flipped RGB8 with a truncating *255 (:323-339).
// --display (ImGui window) is not available on a headless GPU box and is rejected.
// Beyond the reference: the PLY is de-interleaved and activated on the device
// (--ingest device, default; `host` = the reference's read_gs_ply order of work), and --cameras <file> renders a
// batch of views of the resident scene (one `px py pz  tx ty tz  ux uy uz [fov]` line per camera, `#` comments),
// writing <stem>_<backend>_<k>.png per view.  --gpus N shards the views of --cameras over N GPUs, one process per GPU
// (view k on rank k mod N; every rank holds the whole scene; no collective in the forward), and --backward adds, per
// round of N views, each rank's backward (dL/dimg = 1) and the RCCL sum of the dense per-splat gradients over the ranks
// (lcgs_grads_allreduce), printing the norms of the summed gradients -- the multi-view batch of SURVEY 8e driven from C++.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <algorithm>
#include <string>
#include <vector>

#include "lcgs/lcgs.hpp"

namespace
{
[[noreturn]] void die(const std::string& m)
{
    fprintf(stderr, "lcgs-app: %s\n", m.c_str());
    exit(1);
}

void usage(const char* argv0)
{
    printf("Usage: %s [options]\n", argv0);
    printf("  --help / -h              Show this help message\n");
    printf("  --res <width>x<height>   Set the resolution (default: 1600x1063)\n");
    printf("  --ply <path>             Set the path to the PLY file (default: gsplat.ply)\n");
    printf("  --backend <name>         Accepted for compatibility; the only backend is `hip` (gfx950)\n");
    printf("  --out <dir>              Set the output directory (default: out)\n");
    printf("  --world <type>           colmap or blender (default: colmap)\n");
    printf("  --exp_N <N>              Number of frames to render (default: 1)\n");
    printf("  --path <fused|stage|deferred>  One-submission fused frame (default), the three stage-level operators, or the same\n"
           "                           three calls in deferred mode (lcgs_set_stage_mode: the splatter renders the fused frame)\n");
    printf("  --synth <kind:count:seed> Render a synthetic stand-in scene instead of --ply (kind 0 object, 1 unbounded)\n");
    printf("  --ingest <device|host>   De-interleave/activate the PLY on the GPU (default) or on the host\n");
    printf("  --order <file|spatial>   Keep the splats in file order, or re-order them along a Morton curve at load (same image;\n"
           "                           default: spatial for a PLY ingested on the device -- the library's default --, else file)\n");
    printf("  --pose <garden|lego>     The look-at compiled into the reference (garden, app/main.cpp:191-193; default) or the\n"
           "                           alternative it keeps in a comment for lego / bicycle (app/main.cpp:195-197)\n");
    printf("  --cameras <file>         Render every camera of the file: `px py pz tx ty tz ux uy uz [fov]` per line\n");
    printf("  --gpus <N>               Shard the views of --cameras over N GPUs (one process per GPU; default 1)\n");
    printf("  --backward               Per round of views: backward (dL/dimg = 1) + RCCL sum of the gradients over the GPUs\n");
    printf("  --owner                  With --backward: the splat-ownership step instead (every GPU owns P / N rows; 48-byte\n"
           "                           records and 2-D gradients of on-screen rows travel by RCCL send / recv, no dense sum).\n"
           "                           The printed norms come from a verification all-reduce outside the step\n");
    printf("  --comm-selftest          With --backward: before anything else, the communicator's self-test on every rank (1 KB\n"
           "                           all-reduce, zero- and one-byte messages to every peer, an ownership step on a scratch\n"
           "                           scene; 30 s per phase).  A failure is printed and ends the run with a non-zero status\n");
    printf("  --fit <K>                Training without a Python binding (doc/roadmap.md:4), as a demonstration: the loaded scene's\n"
           "                           frame is the target, opacities and base colours are perturbed, K optimiser steps\n"
           "                           (forward, L2 loss, backward, Adam) pull them back; prints the loss per step.  With\n"
           "                           --cameras every step covers all views of the file (lcgs_fit_views)\n");
    printf("  --fused-adam             With --fit on one view: the optimiser is applied inside the backward's per-splat kernel\n"
           "                           (lcgs_render_backward_adam: on-screen splats only, no gradient arrays)\n");
    printf("  --display                Not supported (headless)\n");
}

template <typename T>
lcgs::Buffer<T> upload(const T* h, size_t n)
{
    lcgs::Buffer<T> b(n);
    if (n && hipMemcpy(b.data(), h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) die("H2D copy failed");
    return b;
}
} // namespace

int main(int argc, char** argv)
{
    unsigned    W = 1600, H = 1063; // app/main.cpp:38
    std::string ply_path = "gsplat.ply", backend = "hip", out_dir = "out", world = "colmap", path = "fused", synth;
    std::string ingest = "device", cameras_file, order = "auto", pose = "garden";
    int         exp_N = 1, gpus = 1, fit_steps = 0;
    bool        backward = false, fused_adam = false, owner = false, comm_selftest = false;
    // parse_command (app/command_parser.hpp:5-79): strip leading dashes, `key=value` or `key value`
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i];
        size_t      k   = arg.find_first_not_of('-');
        if (k == std::string::npos || k == 0) die("invalid argument: " + arg);
        std::string kv = arg.substr(k), key = kv, value;
        size_t      eq = kv.find('=');
        if (eq != std::string::npos) {
            key   = kv.substr(0, eq);
            value = kv.substr(eq + 1);
        } else if (i + 1 < argc) {
            std::string next = argv[i + 1];
            bool        flag = !next.empty() && next[0] == '-' && !(next.size() >= 2 && isdigit((unsigned char)next[1]));
            if (!flag) {
                value = next;
                ++i;
            }
        }
        if (key == "help" || key == "h") {
            usage(argv[0]);
            return 0;
        } else if (key == "res") {
            size_t x = value.find('x');
            if (x == std::string::npos) die("Invalid resolution format: '" + value + "'. Expected <width>x<height>");
            W = (unsigned)std::stoi(value.substr(0, x));
            H = (unsigned)std::stoi(value.substr(x + 1));
        } else if (key == "ply") ply_path = value;
        else if (key == "backend") backend = value;
        else if (key == "out") out_dir = value;
        else if (key == "world") {
            if (value == "colmap" || value.empty()) world = "colmap";
            else if (value == "blender") world = "blender";
            else die("Invalid world type: " + value);
        } else if (key == "exp_N") {
            if (value.empty()) die("--exp_N requires a value");
            exp_N = std::stoi(value);
        } else if (key == "path") path = value;
        else if (key == "synth") synth = value;
        else if (key == "ingest") {
            if (value != "device" && value != "host") die("Invalid ingest mode: " + value);
            ingest = value;
        } else if (key == "order") {
            if (value != "file" && value != "spatial") die("Invalid splat order: " + value);
            order = value;
        } else if (key == "pose") {
            if (value != "garden" && value != "lego" && value != "bicycle") die("Invalid pose: " + value);
            pose = value == "garden" ? "garden" : "lego";
        } else if (key == "cameras") cameras_file = value;
        else if (key == "gpus") {
            if (value.empty()) die("--gpus requires a value");
            gpus = std::stoi(value);
            if (gpus < 1 || gpus > 64) die("--gpus out of range");
        } else if (key == "backward") backward = true;
        else if (key == "owner") owner = true;
        else if (key == "comm-selftest" || key == "comm_selftest") comm_selftest = true;
        else if (key == "fit") {
            if (value.empty()) die("--fit requires a value");
            fit_steps = std::stoi(value);
            if (fit_steps < 1 || fit_steps > 100000) die("--fit out of range");
        }
        else if (key == "fused-adam" || key == "fused_adam") fused_adam = true;
        else if (key == "display") die("--display needs a GUI; this build is headless");
        else die("unknown option --" + key);
    }
    // ply stem (app/main.cpp:126-148)
    std::string ply_name = ply_path;
    size_t      sp       = ply_name.find_last_of("/\\");
    if (sp != std::string::npos) ply_name = ply_name.substr(sp + 1);
    size_t ep = ply_name.find_last_of('.');
    if (ep != std::string::npos) ply_name = ply_name.substr(0, ep);
    mkdir(out_dir.c_str(), 0755);

    // ---- one process per GPU: the launcher forks the ranks BEFORE anything touches the GPU and hands every rank > 0 a
    // pipe on which rank 0 will send the communicator's rendezvous token
    int rank = 0;
    std::vector<int> token_write; // rank 0: write ends towards ranks 1 .. N-1
    int              token_read = -1;
    if (gpus > 1) {
        if (cameras_file.empty()) die("--gpus N shards the views of a --cameras file");
        if (path != "fused") die("--gpus N uses the fused frame (--path fused)");
        std::vector<int> rd(gpus, -1), wr(gpus, -1);
        for (int r = 1; r < gpus; ++r) {
            int fds[2];
            if (pipe(fds) != 0) die("pipe() failed");
            rd[r] = fds[0];
            wr[r] = fds[1];
        }
        std::vector<pid_t> kids;
        for (int r = 0; r < gpus; ++r) {
            fflush(stdout);
            fflush(stderr);
            pid_t pid = fork();
            if (pid < 0) die("fork() failed");
            if (pid == 0) {
                rank = r;
                for (int q = 1; q < gpus; ++q) {
                    if (r == 0) {
                        close(rd[q]);
                        token_write.push_back(wr[q]);
                    } else {
                        close(wr[q]);
                        if (q == r) token_read = rd[q];
                        else close(rd[q]);
                    }
                }
                kids.clear();
                break;
            }
            kids.push_back(pid);
        }
        if (!kids.empty()) { // the launcher: wait for the ranks, exit with the first failure
            for (int q = 1; q < gpus; ++q) {
                close(rd[q]);
                close(wr[q]);
            }
            int rc = 0;
            for (pid_t k : kids) {
                int st = 0;
                if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1;
            }
            return rc;
        }
    }
    const bool root = rank == 0;

    try {
        lcgs::Device device(rank);
        // ---- scene: read_gs_ply (app/main.cpp:165-167) or a synthetic stand-in
        lcgs_scene_host sc{};
        std::vector<float> spos, sfeat, sop, sscale, srot;
        int                P = 0;
        bool               on_device = false;
        auto               t_load = std::chrono::steady_clock::now();
        if (!synth.empty()) {
            int kind = 0; long long count = 0; unsigned long long seed = 0;
            if (sscanf(synth.c_str(), "%d:%lld:%llu", &kind, &count, &seed) != 3) die("--synth expects kind:count:seed");
            P = (int)count;
            spos.resize((size_t)P * 3); sfeat.resize((size_t)P * 48); sop.resize(P); sscale.resize((size_t)P * 3); srot.resize((size_t)P * 4);
            lcgs::check(lcgs_synth_scene(kind, seed, 0, count, spos.data(), sfeat.data(), sop.data(), sscale.data(), srot.data()));
            sc = { P, 3, spos.data(), sfeat.data(), sop.data(), sscale.data(), srot.data() };
            ply_name = "synth" + std::to_string(kind) + "_" + std::to_string(count);
        } else if (ingest == "device") {
            // records -> GPU -> activated arrays; the context keeps its scene in spatial order unless --order file
            lcgs::check(lcgs_set_ingest_order(device.ctx(), order == "file" ? LCGS_ORDER_FILE : LCGS_ORDER_SPATIAL));
            lcgs::check(lcgs_scene_load_ply(device.ctx(), ply_path.c_str(), &P));
            on_device = true;
            if (order != "file") order = "done";
        } else {
            lcgs::check(lcgs_ply_read(ply_path.c_str(), &sc));
            P = sc.num_gaussians;
        }
        if (root) printf("num_gaussians: %d\n", P);
        // the five device arrays of app/main.cpp:180-186, 216-223 (owned here, or by the context after a device ingest)
        lcgs::Buffer<float>     o_pos, o_scale, o_rotq, o_sh, o_opacity;
        lcgs::BufferView<float> d_pos, d_scale, d_rotq, d_sh, d_opacity;
        if (on_device) {
            const float *pp, *ps, *pr, *pf, *po;
            lcgs::check(lcgs_scene_pointers(device.ctx(), nullptr, nullptr, &pp, &ps, &pr, &pf, &po));
            d_pos     = { const_cast<float*>(pp), (size_t)P * 3 };
            d_scale   = { const_cast<float*>(ps), (size_t)P * 3 };
            d_rotq    = { const_cast<float*>(pr), (size_t)P * 4 };
            d_sh      = { const_cast<float*>(pf), (size_t)P * 48 };
            d_opacity = { const_cast<float*>(po), (size_t)P };
        } else {
            o_pos = upload(sc.pos, (size_t)P * 3); o_scale = upload(sc.scale, (size_t)P * 3);
            o_rotq = upload(sc.rotq, (size_t)P * 4); o_sh = upload(sc.feature, (size_t)P * 48);
            o_opacity = upload(sc.opacity, (size_t)P);
            d_pos = o_pos; d_scale = o_scale; d_rotq = o_rotq; d_sh = o_sh; d_opacity = o_opacity;
            lcgs::check(lcgs_scene_bind(device.ctx(), P, 3, d_pos.ptr, d_scale.ptr, d_rotq.ptr, d_sh.ptr, d_opacity.ptr));
        }
        if (order == "spatial" && P > 0) { // lcgs_scene_reorder_spatial: same image, the splats of a view in runs of rows
            const float *pp, *ps, *pr, *pf, *po;
            lcgs::check(lcgs_scene_reorder_spatial(device.ctx(), nullptr));
            lcgs::check(lcgs_scene_pointers(device.ctx(), nullptr, nullptr, &pp, &ps, &pr, &pf, &po));
            d_pos     = { const_cast<float*>(pp), (size_t)P * 3 };
            d_scale   = { const_cast<float*>(ps), (size_t)P * 3 };
            d_rotq    = { const_cast<float*>(pr), (size_t)P * 4 };
            d_sh      = { const_cast<float*>(pf), (size_t)P * 48 };
            d_opacity = { const_cast<float*>(po), (size_t)P };
        }
        if (root) printf("scene resident in %.1f ms (%s ingest)\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load).count(),
               !synth.empty() ? "synthetic" : ingest.c_str());

        // ---- cameras: the hard-coded look-at of app/main.cpp:191-207, or one per line of --cameras
        struct View { float pos[3], target[3], up[3], fov; };
        std::vector<View> views;
        if (cameras_file.empty()) {
            View v = { { -3.0f, -0.5f, 3.3f }, { 0.0f, 3.0f, 0.5f }, { 0.0f, -1.0f, -1.0f }, 0.0f }; // app/main.cpp:191-193
            if (pose == "lego") v = { { -3.0f, -0.5f, 2.3f }, { 0.0f, 0.0f, 0.5f }, { 0.0f, -1.0f, 0.0f }, 0.0f }; // :195-197
            if (world == "blender") { v.up[0] = 0.0f; v.up[1] = 0.0f; v.up[2] = 1.0f; }
            views.push_back(v);
        } else {
            FILE* cf = fopen(cameras_file.c_str(), "r");
            if (!cf) die("cannot open " + cameras_file);
            char line[1024];
            while (fgets(line, sizeof(line), cf)) {
                const char* q = line;
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '#' || *q == '\n' || *q == '\r' || *q == 0) continue;
                View v{};
                int  n = sscanf(q, "%f %f %f %f %f %f %f %f %f %f", &v.pos[0], &v.pos[1], &v.pos[2], &v.target[0], &v.target[1],
                                &v.target[2], &v.up[0], &v.up[1], &v.up[2], &v.fov);
                if (n < 9) die("bad camera line in " + cameras_file + ": " + line);
                if (n < 10) v.fov = 0.0f;
                views.push_back(v);
            }
            fclose(cf);
            if (views.empty()) die("no cameras in " + cameras_file);
        }
        const float bg[3]  = { 0.0f, 0.0f, 0.0f };
        lcgs::Buffer<float> d_img((size_t)W * H * 3);
        lcgs::Buffer<int>   d_radii((size_t)P);
        // stage-level operators and their buffers (the reference's own call sequence, app/main.cpp:227-308)
        lcgs::SHProcessor sh_processor; lcgs::GSProjector projector; lcgs::GSTileSplatter tile_splatter;
        lcgs::Buffer<float> d_color, d_means_2d, d_depth, d_covs_2d;
        lcgs::Buffer<uint32_t> d_tiles, d_offsets, d_lu, d_ls, d_ranges;
        lcgs::Buffer<uint64_t> d_ku, d_ks;
        if (path != "fused" && path != "stage" && path != "deferred") die("Invalid path: " + path);
        const bool stage_calls = path == "stage" || path == "deferred"; // app/main.cpp's own three calls
        if (stage_calls) {
            sh_processor.create(device); projector.create(device); tile_splatter.create(device);
            if (path == "deferred") lcgs::check(lcgs_set_stage_mode(device.ctx(), LCGS_STAGES_DEFERRED));
            d_color = lcgs::Buffer<float>((size_t)P * 3); d_means_2d = lcgs::Buffer<float>((size_t)P * 2);
            d_depth = lcgs::Buffer<float>((size_t)P); d_covs_2d = lcgs::Buffer<float>((size_t)P * 3);
            (void)hipMemset(d_means_2d.data(), 0, (size_t)P * 8); (void)hipMemset(d_depth.data(), 0, (size_t)P * 4); (void)hipMemset(d_covs_2d.data(), 0, (size_t)P * 12);
            d_tiles = lcgs::Buffer<uint32_t>((size_t)P); d_offsets = lcgs::Buffer<uint32_t>((size_t)P);
            const size_t L = 20000000; // app/main.cpp:245
            d_ku = lcgs::Buffer<uint64_t>(L); d_ks = lcgs::Buffer<uint64_t>(L);
            d_lu = lcgs::Buffer<uint32_t>(L); d_ls = lcgs::Buffer<uint32_t>(L);
            d_ranges = lcgs::Buffer<uint32_t>((size_t)((W + 15) / 16) * ((H + 15) / 16) * 2);
        }
        std::vector<float>   h_img((size_t)W * H * 3);
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        auto make_camera = [&](const View& v) {
            lcgs::Camera cam = lcgs::get_lookat_cam(v.pos, v.target, v.up);
            if (v.fov > 0.0f) cam.fov = v.fov;
            cam.aspect_ratio = (float)W / (float)H;
            cam.width        = (int)W;
            cam.height       = (int)H;
            return cam;
        };
        auto save_view = [&](const float* d_view, size_t vi) {
            if (hipMemcpy(h_img.data(), d_view, h_img.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) die("D2H copy failed");
            lcgs_image_to_rgb8((int)W, (int)H, h_img.data(), rgb.data());
            std::string img_name = out_dir + "/" + ply_name + "_" + backend +
                                   (cameras_file.empty() ? std::string() : "_" + std::to_string(vi)) + ".png";
            lcgs::check(lcgs_write_png(img_name.c_str(), (int)W, (int)H, rgb.data()));
            printf("result saved in %s\n", img_name.c_str());
        };
        if (fit_steps > 0) {
            // ---- "training without python binding" (doc/roadmap.md:4) through the C ABI only: the scene's own frame is the
            // target; the optimiser starts from perturbed opacities / base colours and has to find its way back.
            if (gpus > 1) die("--fit runs on one GPU");
            const size_t n3 = (size_t)P * 3, n4 = (size_t)P * 4, n48 = (size_t)P * 48;
            std::vector<float> h_pos(n3), h_scale(n3), h_rotq(n4), h_sh(n48), h_op(P);
            lcgs::check(lcgs_scene_download(device.ctx(), h_pos.data(), h_scale.data(), h_rotq.data(), h_sh.data(), h_op.data()));
            // one target per view of --cameras (the scene's own frames): an optimiser step covers all of them
            // (lcgs_fit_views: forward -> L2 loss -> backward per view, gradients summed, consecutive views overlapping)
            lcgs::Camera cam = make_camera(views[0]);
            lcgs::Scene  scene(device);
            std::vector<lcgs::Camera>         cams;
            std::vector<lcgs::Buffer<float>>  targets;
            std::vector<const float*>         target_ptrs;
            for (const auto& v : views) {
                cams.push_back(make_camera(v));
                targets.emplace_back((size_t)W * H * 3);
                scene.render(cams.back(), targets.back(), bg);
                target_ptrs.push_back(targets.back().data());
            }
            const int           nv = (int)cams.size();
            lcgs::Buffer<float> d_loss((size_t)nv);
            std::vector<float>  h_loss((size_t)nv);
            // raw 3DGS parameters (log-scale, opacity logit, un-normalised quaternion) of the PERTURBED scene
            std::vector<float> r_scale(n3), r_op(P);
            for (size_t i = 0; i < n3; ++i) r_scale[i] = std::log(h_scale[i]);
            for (int i = 0; i < P; ++i) {
                const float o = std::min(std::max(h_op[i], 1e-6f), 1.0f - 1e-6f);
                r_op[i]       = std::log(o / (1.0f - o)) - 1.0f; // every splat more transparent than it should be
                h_op[i]       = 1.0f / (1.0f + std::exp(-r_op[i]));
            }
            for (int i = 0; i < P; ++i)
                for (int c = 0; c < 3; ++c) h_sh[(size_t)i * 48 + c] += 0.3f * ((float)((i * 3 + c) % 7) / 3.0f - 1.0f);
            lcgs::Buffer<float> a_pos = upload(h_pos.data(), n3), a_scale = upload(h_scale.data(), n3),
                                a_rotq = upload(h_rotq.data(), n4), a_sh = upload(h_sh.data(), n48), a_op = upload(h_op.data(), (size_t)P);
            lcgs::Buffer<float> w_scale = upload(r_scale.data(), n3), w_rotq = upload(h_rotq.data(), n4), w_op = upload(r_op.data(), (size_t)P);
            auto zeros = [&](size_t n) {
                lcgs::Buffer<float> b(n);
                if (hipMemset(b.data(), 0, n * sizeof(float)) != hipSuccess) die("memset failed");
                return b;
            };
            lcgs::Buffer<float> g[5] = { zeros(n3), zeros(n3), zeros(n4), zeros(n48), zeros((size_t)P) };
            lcgs::Buffer<float> m[5] = { zeros(n3), zeros(n3), zeros(n4), zeros(n48), zeros((size_t)P) };
            lcgs::Buffer<float> v[5] = { zeros(n3), zeros(n3), zeros(n4), zeros(n48), zeros((size_t)P) };
            scene.bind(P, a_pos, a_scale, a_rotq, a_sh, a_op);
            const lcgs_grads  grads = { g[0].data(), g[1].data(), g[2].data(), g[3].data(), g[4].data() };
            const lcgs_params raw = { a_pos.data(), w_scale.data(), w_rotq.data(), a_sh.data(), w_op.data() }; // pos / sh: raw == activated
            const lcgs_params act = { a_pos.data(), a_scale.data(), a_rotq.data(), a_sh.data(), a_op.data() };
            const lcgs_params mm = { m[0].data(), m[1].data(), m[2].data(), m[3].data(), m[4].data() };
            const lcgs_params vv = { v[0].data(), v[1].data(), v[2].data(), v[3].data(), v[4].data() };
            lcgs_adam_config cfg = { 0.0f, 2.5e-2f, 0.0f, 5e-2f, 0.0f, 0.0f, 0.9f, 0.999f, 1e-15f, 1, 0 }; // opacity + dc only
            auto t0 = std::chrono::steady_clock::now();
            float first_loss = 0.0f, loss = 0.0f;
            if (fused_adam && nv != 1) die("--fused-adam differentiates one view per step (omit --cameras)");
            lcgs::Buffer<float> d_dL(fused_adam ? (size_t)W * H * 3 : 1);
            for (int it = 0; it < fit_steps; ++it) {
                cfg.step = it + 1;
                if (fused_adam) { // forward (kept state) -> L2 loss and its gradient -> backward with Adam folded in
                    scene.render(cams[0], d_img, bg, 1.0f, /*keep_state=*/true);
                    lcgs::check(lcgs_l2_loss_backward(device.ctx(), (int)W, (int)H, d_img.data(), target_ptrs[0], d_dL.data(),
                                                      d_loss.data()));
                    lcgs::check(lcgs_render_backward_adam(device.ctx(), d_dL.data(), P, 3, &cfg, &raw, &mm, &vv, &act));
                } else {
                    scene.fit_views(cams, target_ptrs, grads, d_loss.data(), bg);
                    lcgs::check(lcgs_adam_step(device.ctx(), P, 3, &cfg, &grads, &raw, &mm, &vv, &act));
                }
                device.synchronize();
                if (hipMemcpy(h_loss.data(), d_loss.data(), sizeof(float) * nv, hipMemcpyDeviceToHost) != hipSuccess) die("D2H copy failed");
                loss = 0.0f;
                for (float l : h_loss) loss += l / (float)nv; // mean over the views
                if (it == 0) first_loss = loss;
                printf("step %d loss %.6e\n", it + 1, loss);
            }
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("fit: loss %.6e -> %.6e in %d steps of %d view(s) (%.3f ms per step)\n", first_loss, loss, fit_steps, nv,
                   ms / fit_steps);
            scene.render(cam, d_img, bg);
            save_view(d_img.data(), 0);
            views.clear(); // done
        }
        if (gpus > 1 || backward) {
            // ---- a multi-view batch sharded over the GPUs: view k belongs to rank k mod N; one round = N
            // views, one per rank.  Forward: no collective.  --backward: each rank differentiates its view (dL/dimg = 1,
            // i.e. the gradient of the image sum) and the dense per-splat gradients are summed over the ranks by RCCL.
            if (path != "fused") die("--backward uses the fused frame (--path fused)");
            std::unique_ptr<lcgs::Comm> comm;
            if (backward) {
                lcgs_comm_id id{};
                if (root) {
                    id = lcgs::Comm::unique_id();
                    for (int fd : token_write)
                        if (write(fd, id.bytes, sizeof(id.bytes)) != (ssize_t)sizeof(id.bytes)) die("cannot send the communicator token");
                } else if (read(token_read, id.bytes, sizeof(id.bytes)) != (ssize_t)sizeof(id.bytes))
                    die("cannot receive the communicator token");
                comm.reset(new lcgs::Comm(device, id, rank, gpus));
                if (comm_selftest) { // what the communicator says about itself, before it carries anything that matters
                    const lcgs_comm_selftest_report t = comm->selftest(30.0, /*throw_on_failure=*/false);
                    printf("rank %d / %d communicator self-test: all-reduce %s (%.2f ms), point-to-point %s (%.2f ms), ownership step %s "
                           "(%.1f ms, worst gradient error %.1e)%s%s\n",
                           t.rank, t.world_size, t.allreduce_ok == 1 ? "ok" : "WRONG", t.allreduce_ms, t.p2p_ok == 1 ? "ok" : "WRONG",
                           t.p2p_ms, t.owner_step_ok == 1 ? "ok" : (t.owner_step_ok < 0 ? "not run" : "WRONG"), t.owner_step_ms,
                           t.owner_max_grad_err, t.timed_out ? " -- TIMED OUT: " : "", t.timed_out ? t.message : "");
                    fflush(stdout);
                    if (t.timed_out || t.allreduce_ok != 1 || t.p2p_ok != 1 || t.owner_step_ok == 0) {
                        fprintf(stderr, "communicator self-test failed on rank %d: %s\n", rank, t.message);
                        _exit(3); // (a phase may be stuck inside RCCL: no destructors, no second attempt from this process)
                    }
                }
            }
            lcgs::Scene         scene(device);
            lcgs::Buffer<float> g_pos, g_scale, g_rotq, g_sh, g_op, d_ones;
            lcgs_grads          grads{};
            const size_t        widths[5] = { 3, 3, 4, 48, 1 };
            if (backward) {
                g_pos = lcgs::Buffer<float>((size_t)P * 3); g_scale = lcgs::Buffer<float>((size_t)P * 3);
                g_rotq = lcgs::Buffer<float>((size_t)P * 4); g_sh = lcgs::Buffer<float>((size_t)P * 48);
                g_op = lcgs::Buffer<float>((size_t)P);
                grads = { g_pos.data(), g_scale.data(), g_rotq.data(), g_sh.data(), g_op.data() };
                std::vector<float> ones((size_t)W * H * 3, 1.0f);
                d_ones = upload(ones.data(), ones.size());
            }
            const size_t rounds = (views.size() + (size_t)gpus - 1) / (size_t)gpus;
            auto         t0     = std::chrono::steady_clock::now();
            if (owner) {
                if (!backward) die("--owner goes with --backward");
                if (views.size() % (size_t)gpus != 0) die("--owner needs a multiple of --gpus views (every rank renders one per round)");
            }
            for (size_t round = 0; owner && round < rounds; ++round) {
                // ---- splat ownership: this rank owns P / N rows and renders view round * N + rank; records out,
                // 2-D gradients back, over RCCL point-to-point (lcgs_owner_step_forward / _backward)
                std::vector<lcgs::Camera> cams;
                for (int q = 0; q < gpus; ++q) cams.push_back(make_camera(views[round * (size_t)gpus + (size_t)q]));
                float* gp[5] = { g_pos.data(), g_scale.data(), g_rotq.data(), g_sh.data(), g_op.data() };
                for (int a = 0; a < 5; ++a) // rows of other owners stay zero: the verification sum below needs that
                    if (hipMemsetAsync(gp[a], 0, (size_t)P * widths[a] * 4, nullptr) != hipSuccess) die("memset failed");
                if (hipDeviceSynchronize() != hipSuccess) die("sync failed");
                // (from the second round on without a host read-back: sizes from the previous round's counts, one verdict
                // behind the step, repeated by every rank if any rank's was short)
                comm->owner_step_set_async(true);
                const int redone = comm->owner_step(cams.data(), bg, d_img.data(), d_ones.data(), grads);
                if (redone && root) printf("round %zu: the ownership step was repeated %d time(s)\n", round, redone);
                save_view(d_img.data(), round * (size_t)gpus + (size_t)rank);
                device.synchronize();
                const lcgs_comm_stats st = comm->stats();
                comm->allreduce(P, grads); // VERIFICATION only (the step itself never sums dense rows): norms as --backward prints them
                device.synchronize();
                if (root) {
                    static const char* names[5] = { "pos", "scale", "rotq", "sh", "opacity" };
                    printf("round %zu (%d view%s, ownership step: %lld bytes sent by rank 0): grad_l2", round, gpus, gpus > 1 ? "s" : "",
                           (long long)st.bytes_sent);
                    for (int a = 0; a < 5; ++a) {
                        std::vector<float> h((size_t)P * widths[a]);
                        if (hipMemcpy(h.data(), gp[a], h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) die("D2H copy failed");
                        double acc = 0.0;
                        for (float x : h) acc += (double)x * x;
                        printf(" %s %.9e", names[a], std::sqrt(acc));
                    }
                    printf("\n");
                }
            }
            for (size_t round = 0; !owner && round < rounds; ++round) {
                const size_t vi   = round * (size_t)gpus + (size_t)rank;
                const bool   mine = vi < views.size();
                int          n    = 0;
                if (mine) {
                    lcgs::Camera cam = make_camera(views[vi]);
                    n                = scene.render(cam, d_img, bg, 1.0f, backward);
                    save_view(d_img.data(), vi);
                }
                if (!backward) continue;
                float* gp[5] = { g_pos.data(), g_scale.data(), g_rotq.data(), g_sh.data(), g_op.data() };
                if (mine && n > 0) scene.backward(d_ones, grads);
                else // no view for this rank in the last round (or an empty frame): it contributes zeros to the sum
                    for (int a = 0; a < 5; ++a)
                        if (hipMemsetAsync(gp[a], 0, (size_t)P * widths[a] * 4, nullptr) != hipSuccess) die("memset failed");
                if (!(mine && n > 0) && hipDeviceSynchronize() != hipSuccess) die("sync failed");
                comm->allreduce(P, grads);
                device.synchronize();
                if (root) { // the norms of the summed gradients: identical on every rank
                    static const char* names[5] = { "pos", "scale", "rotq", "sh", "opacity" };
                    printf("round %zu (%d view%s): grad_l2", round, (int)std::min<size_t>(gpus, views.size() - round * gpus),
                           gpus > 1 ? "s" : "");
                    for (int a = 0; a < 5; ++a) {
                        std::vector<float> h((size_t)P * widths[a]);
                        if (hipMemcpy(h.data(), gp[a], h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) die("D2H copy failed");
                        double acc = 0.0;
                        for (float x : h) acc += (double)x * x;
                        printf(" %s %.9e", names[a], std::sqrt(acc));
                    }
                    printf("\n");
                }
            }
            device.synchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (root)
                printf("exp time: %.3f ms\n%zu views on %d GPU%s%s\n", ms, views.size(), gpus, gpus > 1 ? "s" : "",
                       backward ? ", gradients summed over the GPUs" : "");
            views.clear(); // done
        }
        if (!cameras_file.empty() && path == "fused" && exp_N == 1 && !views.empty()) {
            // the whole camera file as one batch: two frames in flight (lcgs_render_forward_batch)
            lcgs::Scene               scene(device);
            std::vector<lcgs::Camera> cams;
            std::vector<lcgs::Buffer<float>> imgs;
            std::vector<float*>       img_ptrs;
            for (const View& v : views) {
                cams.push_back(make_camera(v));
                imgs.emplace_back((size_t)W * H * 3);
                img_ptrs.push_back(imgs.back().data());
            }
            auto t0 = std::chrono::steady_clock::now();
            scene.render_batch(cams, img_ptrs, bg);
            device.synchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("exp time: %.3f ms\nfps: %.2f with %zu views in one batch\n", ms, 1000.0 * views.size() / ms, views.size());
            for (size_t vi = 0; vi < views.size(); ++vi) save_view(img_ptrs[vi], vi);
            views.clear(); // done
        }
        for (size_t vi = 0; vi < views.size(); ++vi) {
            lcgs::Camera cam = make_camera(views[vi]);
            int  num_rendered = 0;
            auto t0           = std::chrono::steady_clock::now();
            if (stage_calls) {
                for (int it = 0; it < exp_N; ++it) {
                    sh_processor.process({ P, 3, d_pos }, cam, d_sh, d_color, 3, 3);
                    projector.forward({ P, d_pos, d_scale, d_rotq, 1.0f }, { d_means_2d, d_covs_2d, d_depth }, cam);
                    num_rendered = tile_splatter.forward({ d_tiles, d_offsets, d_ku, d_lu, d_ks, d_ls, d_ranges },
                                                         { P, { 0, 0, 0 }, d_means_2d, d_depth, d_covs_2d, d_color, d_opacity },
                                                         { (int)H, (int)W, d_img, d_radii });
                }
            } else {
                for (int it = 0; it < exp_N; ++it)
                    lcgs::check(lcgs_render_forward(device.ctx(), &cam, bg, 1.0f, d_img.data(), d_radii.data(), 0,
                                                    it + 1 == exp_N ? &num_rendered : nullptr));
            }
            device.synchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("num_rendered: %d\nexp time: %.3f ms\nfps: %.2f with test N %d\n", num_rendered, ms, 1000.0 / (ms / exp_N), exp_N);
            save_view(d_img.data(), vi);
        }
        if (synth.empty() && !on_device) lcgs_scene_host_free(&sc);
    } catch (const lcgs::Error& e) {
        die(e.what());
    }
    return 0;
}
