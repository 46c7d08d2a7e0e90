// rt_world.h — host-side World / WorldBuilder / Camera mirror (C++), internal header.
//
// Mirrors the reference's host API (world.rs:15-99, object_store.rs:185-241,360-389,
// camera.rs:10-37, geometry.rs:12-67) closely enough that scenes are built by the
// same sequence of calls, but stores the graph as one unified node table that the
// flattener can hand to rtx_scene_create() directly.
#ifndef RT_WORLD_H
#define RT_WORLD_H
#include <string>
#include <vector>

#include "../../include/rtx_host.h"

namespace rth {

// 4x4 f64, column-major like nalgebra's Matrix4 storage (e[c*4+r]).
struct Mat4 {
    double e[16];
    double& operator()(int r, int c) { return e[c * 4 + r]; }
    double operator()(int r, int c) const { return e[c * 4 + r]; }
};

Mat4 mat_identity();
Mat4 mat_translation(double x, double y, double z);      // geometry.rs:16-23
Mat4 mat_scaling(double x, double y, double z);          // geometry.rs:25-31
Mat4 mat_rotation(int axis, double angle);               // geometry.rs:33-47 (Rotation3::from_axis_angle)
Mat4 mat_view(const double* from, const double* to, const double* up);   // geometry.rs:53-67
Mat4 mat_mul(const Mat4& a, const Mat4& b);              // nalgebra gemm order
Mat4 mat_inverse(const Mat4& a);                         // nalgebra do_inverse4
void mat_point(const Mat4& m, const double* p, double* out);   // Projective3 * Point3

enum ObjKind { OBJ_GROUP = 0, OBJ_CSG = 1, OBJ_SHAPE = 2 };

struct Node {
    int kind;                 // RTX_NODE_*
    int parent = -1;          // node id
    Mat4 transform;
    std::vector<int> children;   // groups: insertion order; CSG: {left, right}
    int csg_op = 0;
    int shape_index = -1;
    rtxh_material material;   // shapes (by value, shapes/mod.rs:80-82)
    double bounds[6];         // groups/CSGs: local bounds, NaN = empty (bounds.rs:23-27)
};

struct Light { double pos[3]; double intensity[3]; };

struct World {
    std::vector<Node> nodes;
    std::vector<int> shape_nodes, group_nodes, csg_nodes;   // per-kind index -> node id
    std::vector<int> scene;                                 // root node ids (world.rs:82)
    std::vector<Light> lights;

    int node_of(int kind, int index) const;
    int add_shape(int type, const rtxh_material* m, const Mat4& t);       // object_store.rs:211-217
    int add_group(const Mat4& t);                                         // :219-223
    void set_group_of(int kind, int index, int group);                    // :238-241
    int add_csg(int op, int lk, int li, int rk, int ri, const Mat4& t);   // :225-236
    void add_object_to_scene(int kind, int index);                        // world.rs:67-69
    int add_light(const double* pos, const double* intensity);            // world.rs:71-74
    void build();                                                         // world.rs:30-43
    void set_bounds_of(int node, double* out6);                           // object_store.rs:360-389
};

// Owns the arrays an rtx_scene_desc points into.
struct Flattened {
    std::vector<int32_t> kind, first_child, child_count, csg_op, shape_index, material, children, roots;
    std::vector<double> inverse, bounds, light_pos, light_int;
    std::vector<rtx_material> materials;
    rtx_scene_desc desc;
};
void flatten(const World& w, Flattened& out);

void default_material(rtxh_material* m);
void camera_new(int hsize, int vsize, double fov, rtxh_camera* c);        // camera.rs:20-37

// yaml.rs:22-73 (yaml_loader.cpp)
bool load_yaml(const std::string& text, World& world, std::vector<rtxh_camera>& cameras, std::string& err);
// canvas.rs:39-53 (png_writer.cpp)
bool write_png(const std::string& path, const uint8_t* rgb, int w, int h, std::string& err);

}  // namespace rth

struct rtxh_world { rth::World w; };

#endif
