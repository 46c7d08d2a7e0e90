/* rtx_host.h — C ABI of the host-side mirror of the reference's public Rust API
 * (exported by the same librtx_hip.so).
 *
 * north_star keeps scene loading, the camera/world/material API and PNG output on
 * the (Rust) host.  No Rust toolchain exists in the build image, so the host side
 * is C++ and is exposed with C linkage so that tests (ctypes), the CLI and other
 * FFI users can drive it.  Names and argument meaning follow the reference:
 *
 *   rtxh_world_new .. rtxh_build        WorldBuilder            world.rs:15-79
 *   rtxh_camera_new                     Camera::new             camera.rs:20-37
 *   rtxh_render                         Camera::render          camera.rs:39-54
 *   rtxh_identity .. rtxh_view_transform  geometry.rs:12-67
 *   rtxh_default_material               Material::new           material.rs:17-22,41-51
 *   rtxh_load_yaml                      load_world_and_cameras_from_str   yaml.rs:22-73
 *   rtxh_write_png                      Canvas::write_image     canvas.rs:39-53
 *   rtxh_render_file                    main.rs:20-28
 *
 * Object references are (kind, index) pairs, kind 0=Group 1=CSG 2=Shape
 * (ObjectIndex, object_store.rs:17-21).  Matrices are 16 doubles, column-major.
 * rtxh_render has NO CPU path: it flattens the world (inverses, bounds, CSG
 * trees, pattern parameters) and calls rtx_scene_create/rtx_render (rtx.h).
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H
#include "rtx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Material as the host API sees it: pattern transform NOT inverted (patterns.rs:9-12) */
typedef struct rtxh_material {
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double color_a[3], color_b[3];
    double pattern_transform[16];
    int32_t pattern_kind;
    int32_t _pad;
} rtxh_material;

/* Camera (camera.rs:10-17) */
typedef struct rtxh_camera {
    int32_t hsize, vsize;
    double pixel_size, half_width, half_height;
    double transform[16];
} rtxh_camera;

typedef struct rtxh_world rtxh_world;

/* geometry.rs */
void rtxh_identity(double* out16);
void rtxh_translation(double x, double y, double z, double* out16);
void rtxh_scaling(double x, double y, double z, double* out16);
void rtxh_rotation_x(double a, double* out16);
void rtxh_rotation_y(double a, double* out16);
void rtxh_rotation_z(double a, double* out16);
void rtxh_view_transform(const double* from3, const double* to3, const double* up3, double* out16);
void rtxh_mat_mul(const double* a16, const double* b16, double* out16);
void rtxh_inverse(const double* m16, double* out16);
void rtxh_default_material(rtxh_material* out);

/* WorldBuilder (world.rs:15-79); builder and built World share the handle */
rtxh_world* rtxh_world_new(void);
void rtxh_world_free(rtxh_world*);
int  rtxh_add_shape(rtxh_world*, int shape_type, const rtxh_material*, const double* transform16);
int  rtxh_add_group(rtxh_world*, const double* transform16);
void rtxh_set_group_of(rtxh_world*, int obj_kind, int obj_index, int group);
int  rtxh_add_csg(rtxh_world*, int op, int lkind, int lindex, int rkind, int rindex, const double* transform16);
void rtxh_add_object_to_scene(rtxh_world*, int obj_kind, int obj_index);
int  rtxh_add_light(rtxh_world*, const double* pos3, const double* intensity3);
void rtxh_build(rtxh_world*);                      /* compute_bounds (world.rs:39-43) */
void rtxh_set_material_of(rtxh_world*, int shape, const rtxh_material*);
void rtxh_get_material_of(rtxh_world*, int shape, rtxh_material* out);
void rtxh_set_light(rtxh_world*, int i, const double* pos3, const double* intensity3);
void rtxh_get_bounds(rtxh_world*, int obj_kind, int obj_index, double* out6);

void rtxh_camera_new(int hsize, int vsize, double fov, rtxh_camera* out);

/* Canonical byte dump of a world (nodes in creation order: kind, parent, transform, children,
 * csg op, shape index, material, bounds; then scene roots and lights).  Returns the size needed;
 * writes only if cap is large enough.  Used to compare two loaders' results bit for bit. */
int64_t rtxh_world_serialize(const rtxh_world*, uint8_t* buf, int64_t cap);

/* Flatten + upload: keeps the scene resident for repeated renders. */
int  rtxh_scene_create(rtx_ctx*, rtxh_world*, rtx_scene** out);
/* rtxh_camera -> rtx_camera (inverts the transform once, camera.rs:65-66) */
void rtxh_camera_flatten(const rtxh_camera*, rtx_camera* out);
/* Camera::render: flatten, upload, render all bands, free.  rgb: hsize*vsize*3, hit may be NULL. */
int  rtxh_render(rtx_ctx*, rtxh_world*, const rtxh_camera*, int max_depth, uint32_t flags,
                 uint8_t* rgb, int32_t* hit, rtx_stats* stats);

/* yaml.rs:22-73.  Returns 0 or RTX_E_INVALID (message via rtxh_last_error()).  Cameras are
 * returned in document order; the CLI renders cameras[0] only (main.rs:24). */
int  rtxh_load_yaml(const char* text, rtxh_world** world_out, rtxh_camera* cameras_out, int max_cameras,
                    int* n_cameras_out);
const char* rtxh_last_error(void);
/* canvas.rs:39-53: 8-bit RGB PNG */
int  rtxh_write_png(const char* path, const uint8_t* rgb, int width, int height);
/* main.rs:20-28: load scene file, render cameras[0] on `device`, write <stem>.png into out_dir */
int  rtxh_render_file(const char* yaml_path, const char* out_dir, int device);
/* The same on n_gpus GPUs of this node (0 = all visible; one process, one context per GPU): cameras[0] is cut into
 * row-interleaved tile bands, one band set per GPU, gathered with RCCL (rtx_render_multi in rtx.h). */
int  rtxh_render_file_gpus(const char* yaml_path, const char* out_dir, int n_gpus);
/* N4: the same for EVERY camera of the file through rtx_render_batch (scene uploaded once): cameras[0] ->
 * <stem>.png as above, cameras[i>0] -> <stem>.cam<i>.png.  n_rendered_out (may be NULL) = images written. */
int  rtxh_render_file_all(const char* yaml_path, const char* out_dir, int device, int* n_rendered_out);

#ifdef __cplusplus
}
#endif
#endif
