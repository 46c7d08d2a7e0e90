// rt_frontend.cpp — rtxh_load_yaml / rtxh_write_png / rtxh_render_file: the C entry points for
// yaml.rs:22-73, canvas.rs:39-53 and main.rs:20-28.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rt_world.h"

static thread_local std::string g_err;

extern "C" {

const char* rtxh_last_error(void) { return g_err.c_str(); }

int rtxh_load_yaml(const char* text, rtxh_world** world_out, rtxh_camera* cameras_out, int max_cameras,
                   int* n_cameras_out) {
    if (!text || !world_out) return RTX_E_INVALID;
    *world_out = nullptr;
    rtxh_world* w = new rtxh_world();
    std::vector<rtxh_camera> cams;
    if (!rth::load_yaml(text, w->w, cams, g_err)) { delete w; return RTX_E_INVALID; }
    if (n_cameras_out) *n_cameras_out = (int)cams.size();
    for (int i = 0; i < (int)cams.size() && i < max_cameras && cameras_out; i++) cameras_out[i] = cams[i];
    *world_out = w;
    return RTX_OK;
}

int rtxh_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path) return RTX_E_INVALID;
    return rth::write_png(path, rgb, width, height, g_err) ? RTX_OK : RTX_E_INVALID;
}

int rtxh_render_file(const char* yaml_path, const char* out_dir, int device) {
    if (!yaml_path) return RTX_E_INVALID;
    std::ifstream in(yaml_path, std::ios::binary);
    if (!in) { g_err = std::string("cannot read ") + yaml_path; return RTX_E_INVALID; }   // main.rs:21 unwrap()
    std::stringstream ss; ss << in.rdbuf();
    rtxh_world* w = nullptr; rtxh_camera cam; int ncam = 0;
    int rc = rtxh_load_yaml(ss.str().c_str(), &w, &cam, 1, &ncam);
    if (rc != RTX_OK) return rc;
    if (ncam < 1) { g_err = "scene has no camera"; rtxh_world_free(w); return RTX_E_INVALID; }   // cameras[0] (main.rs:24)
    rtx_ctx* ctx = nullptr;
    rc = rtx_init(device, &ctx);
    if (rc != RTX_OK) { g_err = "rtx_init failed: no usable HIP device"; rtxh_world_free(w); return rc; }
    std::vector<uint8_t> rgb((size_t)cam.hsize * cam.vsize * 3);
    rc = rtxh_render(ctx, w, &cam, 5 /* camera.rs:42 */, 0, rgb.data(), nullptr, nullptr);
    if (rc != RTX_OK) g_err = rtx_last_error(ctx);
    rtx_shutdown(ctx); rtxh_world_free(w);
    if (rc != RTX_OK) return rc;
    // main.rs:25-26: file_name() with extension png, in the current (or given) directory
    std::string p(yaml_path);
    size_t sl = p.find_last_of('/');
    std::string name = sl == std::string::npos ? p : p.substr(sl + 1);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
    std::string out = (out_dir && *out_dir) ? std::string(out_dir) + "/" + name + ".png" : name + ".png";
    return rtxh_write_png(out.c_str(), rgb.data(), cam.hsize, cam.vsize);
}

// main.rs:20-28 with cameras[0] rendered on n_gpus GPUs of this node (0 = all visible): row-interleaved tile bands,
// one band set per GPU, gathered with RCCL (rtx_render_multi)
int rtxh_render_file_gpus(const char* yaml_path, const char* out_dir, int n_gpus) {
    if (!yaml_path || n_gpus < 0) return RTX_E_INVALID;
    std::ifstream in(yaml_path, std::ios::binary);
    if (!in) { g_err = std::string("cannot read ") + yaml_path; return RTX_E_INVALID; }
    std::stringstream ss; ss << in.rdbuf();
    rtxh_world* w = nullptr; rtxh_camera cam; int ncam = 0;
    int rc = rtxh_load_yaml(ss.str().c_str(), &w, &cam, 1, &ncam);
    if (rc != RTX_OK) return rc;
    if (ncam < 1) { g_err = "scene has no camera"; rtxh_world_free(w); return RTX_E_INVALID; }
    std::vector<rtx_ctx*> ctxs;
    std::vector<rtx_comm*> comms;
    std::vector<rtx_scene*> scenes;
    for (int i = 0; n_gpus == 0 || i < n_gpus; i++) {
        rtx_ctx* c = nullptr;
        rc = rtx_init(i, &c);
        if (rc != RTX_OK) { if (n_gpus == 0 && i > 0) rc = RTX_OK; break; }      // 0 = as many as there are
        ctxs.push_back(c);
    }
    const int n = (int)ctxs.size();
    if (rc != RTX_OK || n == 0) { g_err = "rtx_init failed: fewer usable HIP devices than requested"; if (rc == RTX_OK) rc = RTX_E_DEVICE; }
    if (rc == RTX_OK) {
        comms.assign(n, nullptr);
        rc = rtx_comm_init_all(ctxs.data(), n, comms.data());
        if (rc != RTX_OK) g_err = rtx_last_error(ctxs[0]);
    }
    for (int i = 0; i < n && rc == RTX_OK; i++) {
        rtx_scene* s = nullptr;
        rc = rtxh_scene_create(ctxs[i], w, &s);
        if (rc != RTX_OK) g_err = rtx_last_error(ctxs[i]); else scenes.push_back(s);
    }
    std::vector<uint8_t> rgb((size_t)cam.hsize * cam.vsize * 3);
    if (rc == RTX_OK) {
        rtx_camera flat; rtxh_camera_flatten(&cam, &flat);
        rc = rtx_render_multi(comms.data(), scenes.data(), n, &flat, 5 /* camera.rs:42 */, 0, rgb.data(), nullptr, nullptr);
        if (rc != RTX_OK) g_err = rtx_last_error(ctxs[0]);
    }
    for (rtx_scene* s : scenes) rtx_scene_destroy(s);
    for (rtx_comm* c : comms) rtx_comm_destroy(c);
    for (rtx_ctx* c : ctxs) rtx_shutdown(c);
    rtxh_world_free(w);
    if (rc != RTX_OK) return rc;
    std::string p(yaml_path);
    size_t sl = p.find_last_of('/');
    std::string name = sl == std::string::npos ? p : p.substr(sl + 1);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
    std::string out = (out_dir && *out_dir) ? std::string(out_dir) + "/" + name + ".png" : name + ".png";
    return rtxh_write_png(out.c_str(), rgb.data(), cam.hsize, cam.vsize);
}

int rtxh_render_file_all(const char* yaml_path, const char* out_dir, int device, int* n_rendered_out) {
    if (n_rendered_out) *n_rendered_out = 0;
    if (!yaml_path) return RTX_E_INVALID;
    std::ifstream in(yaml_path, std::ios::binary);
    if (!in) { g_err = std::string("cannot read ") + yaml_path; return RTX_E_INVALID; }
    std::stringstream ss; ss << in.rdbuf();
    const std::string text = ss.str();
    rtxh_world* w = nullptr; int ncam = 0;
    std::vector<rtxh_camera> cams(64);                                       // one parse (warnings print once)
    int rc = rtxh_load_yaml(text.c_str(), &w, cams.data(), (int)cams.size(), &ncam);
    if (rc != RTX_OK) return rc;
    if (ncam < 1) { g_err = "scene has no camera"; rtxh_world_free(w); return RTX_E_INVALID; }
    if (ncam > (int)cams.size()) { g_err = "more than 64 cameras"; rtxh_world_free(w); return RTX_E_CAPACITY; }
    cams.resize(ncam);
    rtx_ctx* ctx = nullptr; rtx_scene* scene = nullptr;
    rc = rtx_init(device, &ctx);
    if (rc != RTX_OK) { g_err = "rtx_init failed: no usable HIP device"; rtxh_world_free(w); return rc; }
    rc = rtxh_scene_create(ctx, w, &scene);
    std::vector<rtx_camera> flat(ncam);
    std::vector<std::vector<uint8_t>> images(ncam);
    std::vector<uint8_t*> ptrs(ncam);
    for (int i = 0; i < ncam && rc == RTX_OK; i++) {
        if (cams[i].hsize <= 0 || cams[i].vsize <= 0) { g_err = "empty camera"; rc = RTX_E_INVALID; break; }
        rtxh_camera_flatten(&cams[i], &flat[i]);
        images[i].resize((size_t)cams[i].hsize * cams[i].vsize * 3);
        ptrs[i] = images[i].data();
    }
    if (rc == RTX_OK) rc = rtx_render_batch(ctx, scene, flat.data(), ncam, 5 /* camera.rs:42 */, 0, ptrs.data(), nullptr, nullptr);
    if (rc != RTX_OK && g_err.empty()) g_err = rtx_last_error(ctx);
    else if (rc != RTX_OK && ctx && *rtx_last_error(ctx)) g_err = rtx_last_error(ctx);
    if (scene) rtx_scene_destroy(scene);
    rtx_shutdown(ctx); rtxh_world_free(w);
    if (rc != RTX_OK) return rc;
    std::string p(yaml_path);
    size_t sl = p.find_last_of('/');
    std::string name = sl == std::string::npos ? p : p.substr(sl + 1);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
    const std::string stem = (out_dir && *out_dir) ? std::string(out_dir) + "/" + name : name;
    for (int i = 0; i < ncam; i++) {
        const std::string out = i == 0 ? stem + ".png" : stem + ".cam" + std::to_string(i) + ".png";
        rc = rtxh_write_png(out.c_str(), images[i].data(), cams[i].hsize, cams[i].vsize);
        if (rc != RTX_OK) return rc;
        if (n_rendered_out) ++*n_rendered_out;
    }
    return RTX_OK;
}

}  // extern "C"
