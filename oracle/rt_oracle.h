/* rt_oracle.h — C API of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a literal, single-threaded-by-default C++ restatement of the
 * reference renderer's trace-and-shade path (felgru/ray-tracer, Rust):
 *   camera.rs:20-69, world.rs:30-43,100-323, object_store.rs:243-389,
 *   group.rs:64-75, csg.rs:73-136, intersections.rs:29-59, shapes/{mod,sphere,plane,cube,bounds}.rs,
 *   rays.rs:17-23, geometry.rs:12-67, material.rs:41-81, patterns.rs:84-160,
 *   color.rs:32-109, canvas.rs:31-53,
 * plus the nalgebra 0.30.1 operation order the reference relies on
 * (Cargo.lock:368-381; restated from its published source).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (ray-tracer_amd/) never links or calls it.
 *
 * Parity status: pinned against the reference's own known-answer tests
 * (tests/test_oracle_kat.py re-asserts every hot-path #[test] of the
 * reference); ulp-level agreement with nalgebra is unpinned (no Rust
 * toolchain here), see DESIGN.md "Oracle".
 *
 * Conventions: 4x4 matrices are 16 doubles, COLUMN-major (nalgebra storage,
 * m[c*4+r]).  Object references are (kind, index) pairs with kind
 * 0=Group 1=CSG 2=Shape (object_store.rs:17-21).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_OBJ_GROUP = 0, ORA_OBJ_CSG = 1, ORA_OBJ_SHAPE = 2 };
enum { ORA_SPHERE = 0, ORA_PLANE = 1, ORA_CUBE = 2 };               /* shapes/mod.rs:84-89 */
enum { ORA_UNION = 0, ORA_INTERSECTION = 1, ORA_DIFFERENCE = 2 };   /* csg.rs:22-26 */
enum { ORA_PAT_PLAIN = 0, ORA_PAT_STRIPES = 1, ORA_PAT_RINGS = 2,
       ORA_PAT_CHECKERS = 3, ORA_PAT_GRADIENT = 4, ORA_PAT_TEST = 5 }; /* patterns.rs:66-73 */

/* material.rs:11-14,30-38 + patterns.rs:9-12 (by value, as the reference stores it) */
typedef struct ora_material {
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double color_a[3], color_b[3];
    double pattern_transform[16];   /* NOT inverted; the oracle inverts per call like the reference */
    int32_t pattern_kind;
    int32_t _pad;
} ora_material;

/* camera.rs:10-17 */
typedef struct ora_camera {
    int32_t hsize, vsize;
    double pixel_size, half_width, half_height;
    double transform[16];
} ora_camera;

/* event counters (what the reference algorithm executes; see DESIGN.md "flops model") */
enum {
    ORA_EV_XFORM = 0,      /* ObjectStore::intersect calls (ray into node space) */
    ORA_EV_SPHERE,         /* sphere local_intersect */
    ORA_EV_PLANE,
    ORA_EV_CUBE,
    ORA_EV_BOUNDS,         /* Bounds::intersects */
    ORA_EV_CSG_ENTRIES,    /* entries walked by filter_intersections */
    ORA_EV_HIT,            /* shade_hit calls */
    ORA_EV_LEVELS,         /* hierarchy levels walked by normal_at (shape + ancestors) */
    ORA_EV_LIGHT,          /* lighting() evaluations */
    ORA_EV_POW,            /* powf calls */
    ORA_EV_REFR,           /* refracted_color evaluations past the early returns */
    ORA_EV_SCHLICK,        /* schlick() calls */
    ORA_EV_COUNT = 16
};

typedef struct ora_stats {
    uint64_t rays_primary, rays_shadow, rays_secondary;
    uint64_t ev[ORA_EV_COUNT];
    uint64_t tie_pixels;       /* pixels flagged as EPSILON-tie */
    double   ms;               /* wall time of the render call */
} ora_stats;

/* PreparedComputations (world.rs:210-222) */
typedef struct ora_comps {
    double t;
    int32_t shape_index;
    int32_t inside;
    double point[3], over_point[3], under_point[3], eyev[3], normalv[3], reflectv[3];
    double n1, n2;
} ora_comps;

/* ---- geometry.rs ---- */
void ora_identity(double* out16);
void ora_translation(double x, double y, double z, double* out16);
void ora_scaling(double x, double y, double z, double* out16);
void ora_rotation_x(double a, double* out16);
void ora_rotation_y(double a, double* out16);
void ora_rotation_z(double a, double* out16);
void ora_view_transform(const double* from3, const double* to3, const double* up3, double* out16);
void ora_mat_mul(const double* a16, const double* b16, double* out16);
void ora_inverse(const double* m16, double* out16);
void ora_mul_point(const double* m16, const double* p3, double* out3);
void ora_mul_vector(const double* m16, const double* v3, double* out3);
void ora_reflect(const double* v3, const double* n3, double* out3);
void ora_default_material(ora_material* out);

/* ---- WorldBuilder (world.rs:15-79) ---- */
typedef struct ora_world ora_world;     /* builder and built world share the handle */
ora_world* ora_world_new(void);
void ora_world_free(ora_world*);
int  ora_add_shape(ora_world*, int shape_type, const ora_material*, const double* transform16);
int  ora_add_group(ora_world*, const double* transform16);
void ora_set_group_of(ora_world*, int obj_kind, int obj_index, int group);
int  ora_add_csg(ora_world*, int op, int lkind, int lindex, int rkind, int rindex, const double* transform16);
void ora_add_object_to_scene(ora_world*, int obj_kind, int obj_index);
int  ora_add_light(ora_world*, const double* pos3, const double* intensity3);
void ora_build(ora_world*);             /* compute_bounds, world.rs:39-43 */
void ora_set_material_of(ora_world*, int shape, const ora_material*);
void ora_get_material_of(ora_world*, int shape, ora_material* out);
void ora_set_light(ora_world*, int i, const double* pos3, const double* intensity3);
void ora_set_bounds_of(ora_world*, int obj_kind, int obj_index, double* out6); /* object_store.rs:360 */
void ora_get_bounds(ora_world*, int obj_kind, int obj_index, double* out6);
int  ora_parent_of(ora_world*, int obj_kind, int obj_index, int* parent_index); /* returns kind, -1 = None */

/* ---- hot-path hooks ---- */
int  ora_local_intersect(int shape_type, const double* o3, const double* d3, double* out_t2);
void ora_local_normal_at(int shape_type, const double* p3, double* out3);
int  ora_bounds_intersects(const double* bounds6, const double* o3, const double* d3);
int  ora_intersection_allowed(int op, int lhit, int inl, int inr);
/* filter_intersections (csg.rs:94-121): xs must be sorted; left = sorted shape ids */
int  ora_filter_intersections(int op, const int32_t* left, int nleft,
                              const double* t, const int32_t* shape, int n,
                              double* out_t, int32_t* out_shape);
int  ora_object_intersect(ora_world*, int obj_kind, int obj_index, const double* o3, const double* d3,
                          double* out_t, int32_t* out_shape, int cap);
int  ora_world_intersect(ora_world*, const double* o3, const double* d3,
                         double* out_t, int32_t* out_shape, int cap);
int  ora_hit(const double* t, int n);     /* index of first t >= 0, or -1 (intersections.rs:37) */
void ora_color_at(ora_world*, const double* o3, const double* d3, int remaining, double* out3);
int  ora_is_shadowed(ora_world*, const double* p3, int light);
void ora_prepare_computations(ora_world*, int hit, const double* o3, const double* d3,
                              const double* t, const int32_t* shape, int n, ora_comps* out);
void ora_shade_hit(ora_world*, const ora_comps*, int remaining, double* out3);
void ora_reflected_color(ora_world*, const ora_comps*, int remaining, double* out3);
void ora_refracted_color(ora_world*, const ora_comps*, int remaining, double* out3);
double ora_schlick(const ora_comps*);
void ora_world_to_object(ora_world*, int obj_kind, int obj_index, const double* p3, double* out3);
void ora_normal_to_world(ora_world*, int obj_kind, int obj_index, const double* n3, double* out3);
void ora_normal_at(ora_world*, int shape, const double* p3, double* out3);
void ora_pattern_color_at(const ora_material*, const double* p3, double* out3); /* local_color_at */
void ora_object_color_at(ora_world*, int shape, const double* p3, double* out3); /* object_store.rs:293 */
void ora_shader_lighting(const ora_material*, const double* color3,
                         const double* light_pos3, const double* light_int3,
                         const double* pos3, const double* eyev3, const double* normalv3,
                         int in_shadow, double* out3);
void ora_object_lighting(ora_world*, int shape, int light, const double* pos3,
                         const double* eyev3, const double* normalv3, int in_shadow, double* out3);
void ora_color_to_rgb(const double* c3, uint8_t* out3);

/* ---- camera.rs ---- */
void ora_camera_new(int hsize, int vsize, double fov, ora_camera* out);
void ora_ray_for_pixel(const ora_camera*, int x, int y, double* o3, double* d3);
/* Camera::render restricted to the pixel list (x[i],y[i]), i<n; pass NULL,NULL,0 for the
 * whole image.  Outputs are indexed by i (or by x+y*hsize for the whole image); any may be
 * NULL.  hit_id = ShapeIndex of the primary hit, -1 = miss.  tie = 1 if any decision on the
 * pixel's ray tree was within 1e-11 (relative) of flipping (EPSILON-tie pixel), bit1 = primary-hit-id tie.
 * threads <= 1 reproduces the reference's single-threaded loop. */
int  ora_render(ora_world*, const ora_camera*, int max_depth,
                const int32_t* xs, const int32_t* ys, int64_t n,
                uint8_t* rgb, int32_t* hit_id, double* color, uint8_t* tie,
                ora_stats* stats, int threads);

#ifdef __cplusplus
}
#endif
#endif
