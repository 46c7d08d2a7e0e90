// rtx_device.h — device-side scene layout shared by the scene compiler
// (rtx_api.hip) and the kernel (rtx_kernel.hip).
//
// The scene graph (object_store.rs / group.rs / csg.rs) is compiled into
//  * a linear TRAVERSAL PROGRAM: the nodes in the reference's depth-first visiting order (world.rs:113,
//    group.rs:70, csg.rs:80-81) with explicit BEGIN/MID/END records for containers.  The op index is the
//    identity of a shape inside the kernel and equals the visiting order, so "smallest (t, op index) wins"
//    reproduces the stable sorts of intersections.rs:53,58 without any list.  The op interpreter walks the
//    ops of a CSG root (wave-uniform op index, records through scalar loads; lanes that failed a nested
//    container's bounds test sit out until the op index reaches their `resume_at`);
//  * CHILD LISTS (ListChunk / ContRec below) for everything outside CSG, which the kernel walks list by list.
#ifndef RTX_DEVICE_H
#define RTX_DEVICE_H
#include <stdint.h>

enum {
    OPK_SPHERE = 0, OPK_PLANE = 1, OPK_CUBE = 2,     // == RTX_NODE_* for shapes
    OPK_GROUP_BEGIN = 3, OPK_GROUP_END = 4,
    OPK_CSG_BEGIN = 5, OPK_CSG_MID = 6, OPK_CSG_END = 7
};
enum {
    OPF_LIST = 1,      // inside a CSG root: hits go to the per-lane list, not the running state
    OPF_DIV = 2,       // inverse's w33 != 1.0: keep the point division (nalgebra Projective3*Point3)
    OPF_ROOT = 4,      // CSG_BEGIN/CSG_END of a CSG root (a CSG with no CSG ancestor)
    OPF_INLINE = 8     // plane outside CSG without point division: never culled, every lane tests it, so the
                       //   exact test runs wave-uniformly in the sweep; a run is all-INLINE or has none
};

// Each op is split into a 32-byte head (all a culled visit touches; the head of the NEXT op is
// prefetched while the current one is processed) and a 192-byte body (exact test only).  Both are
// read through scalar loads (wave-uniform index).
struct __attribute__((aligned(32))) OpHead {   // 32 bytes = one s_load_dwordx8
    uint32_t info;     // kind | flags << 4 | level << 8 | clevel << 16 | nofilter << 24
                       //   kind: OPK_*; flags: OPF_*; level: nesting depth (the ray to transform is the
                       //   local ray of level-1, 0 = world ray); clevel: container level inside a CSG root
    int32_t skip;      // BEGIN ops: index just past the matching END
    int32_t shape;     // CSG ops: operator (shape ops: ShapeIndex, informational; the kernel identifies a
                       //   shape by its OP INDEX and reads HitRec[op] when it shades)
    int32_t run;       // shape ops: ops left in this run of consecutive same-level shapes (1..60), else 0
    float cull[4];     // bounding sphere in PARENT space: centre xyz, radius^2 (inflated); +inf = never cull
};
#define OPH_KIND(i) ((int)((i) & 15u))
#define OPH_FLAGS(i) ((int)(((i) >> 4) & 15u))
#define OPH_LEVEL(i) ((int)(((i) >> 8) & 255u))
#define OPH_CLEVEL(i) ((int)(((i) >> 16) & 255u))
#define OPH_NOFILTER(i) ((int)(((i) >> 24) & 1u))
#define OPH_XF(i) ((int)(((i) >> 25) & 3u))         // containers: XF_* class of the inverse transform
#define OPH_REGRAY(i) ((int)(((i) >> 27) & 1u))     // CSG root BEGIN: every container below it has the identity as its transform, so the
                                                    //   root's local ray is the only one its program ever needs: it stays in registers
                                                    //   and gets no LDS slot (rtx_kernel.hip: interpret)
// class of a container's inverse transform (decided on the host from its exact bits): what entering it does to a ray
enum { XF_GENERAL = 0,     // any affine map: origin and direction change
       XF_TRANSLATE = 1,   // linear part exactly the identity, w33 == 1: the direction is unchanged, origin = p + t
       XF_IDENTITY = 2 };  // ... and t == 0: the ray is unchanged (CSG operands without a transform of their own)   // container bounds unsuitable for the interval filter (NaN or |finite| > 1e100)
struct __attribute__((aligned(128))) OpBody {   // 128 bytes = one L2 line per shape for the per-lane gathers
    double inv[12];    // rows 0..2 of inverse(transform): inv[r*4+c]
    double w33;        // inverse(3,3)
    int32_t div;       // w33 != 1.0 (copy of OPF_DIV, so shading needs no head load)
    int32_t kind;      // copy of the head's kind, so the per-lane exact tests read one record
    double pad[2];
};

// ---- CSG programs: the ops of one CSG root as self-contained records (all sizes multiples of 8 dwords), in program order.
// The kernel copies a root's program into LDS with one vector load per lane and walks it there (rtx_kernel.hip: interpret).
//   [0] info (as OpHead.info)   [1] BEGIN ops: program offset (dwords) just past the matching END
//   [2] shapes: their OP INDEX; CSG_BEGIN/MID/END: the operator   [3] size of this record in dwords
//   [4..7] bounding sphere in parent space (f32 cx, cy, cz, r^2; +inf = never cull)
//   shapes:  [8..31] inverse rows 0..2 (12 doubles), [32..33] w33
//   BEGIN:   [8..19] local bounds (6 doubles); XF_GENERAL: [20..43] inverse rows, [44..45] w33; XF_TRANSLATE: [20..25] the
//            translation column of the inverse; XF_IDENTITY: nothing more
enum { CSGP_HEAD_DWORDS = 8, CSGP_SHAPE_DWORDS = 40, CSGP_BEGIN_GENERAL_DWORDS = 48, CSGP_BEGIN_TRANSLATE_DWORDS = 32,
       CSGP_BEGIN_IDENTITY_DWORDS = 24 };

// ---- child lists: the traversal outside CSG walks the graph list by list, not op by op.
// A LIST is the ordered children of the world or of a group, cut into chunks of <= 32 children.  One sweep
// (packed f32 bounding-sphere pre-test over the chunk's quads) tells every lane which children it may hit;
// shapes are then tested per lane, and only the containers that SOME lane may hit are visited at all.
struct ListChunk {           // 32 bytes = one s_load_dwordx8
    int32_t n;               // children in this chunk (1..32) | CHUNK_TWO_LEVEL: the chunk's quads have bounding spheres in quad_cull[]
    int32_t child_base;      // multiple of 4: list_ops[child_base + k], spheres at list_cull[child_base * 4 ...]
    uint32_t shape_mask;     // children with a per-lane exact test (spheres, cubes, planes with point division)
    uint32_t plane_mask;     // OPF_INLINE planes: wave-uniform exact test
    uint32_t cont_mask;      // groups and CSG roots
    int32_t next;            // next chunk of the same list, or -1
    int32_t op0;             // >= 0: child k is op (op0 + k), no list_ops lookup needed
    int32_t cont_base;       // cont_recs[cont_base + r] = r-th container child of this chunk
};
#define CHUNK_TWO_LEVEL 0x100u
struct __attribute__((aligned(128))) ContRec {   // 256 bytes; the kernel reads the first half with one batch of scalar loads
    // ---- first half (128 bytes): all a visit needs unless the transform is XF_GENERAL
    float cull[4];           // bounding sphere in parent space (same as the sweep's: the tight one, see list_cull_tight)
    int32_t list;            // groups: first chunk of the child list; CSG roots: offset of the program in csg_prog (dwords)
    int32_t flags;           // CONT_* | XF_* << CONT_XF_SHIFT
    double t[3];             // translation column of the inverse (XF_TRANSLATE: origin = p + t); CSG roots: the first
                             //   32-bit word = length of the program in dwords
    double bounds[6];        // local bounds
    uint32_t child_hdr[8];   // groups: copy of the ListChunk `list` points at (the descent needs no second round trip);
    // ---- second half (128 bytes): XF_GENERAL only
    double inv[12];          // rows 0..2 of inverse(transform)
    double w33;
    int32_t op;              // the container's BEGIN op      (host side)
    int32_t skip;            // op index past the container    (host side)
    float cull_loose[4];     // the sphere around the container's bounds (read in the reference-faithful counting mode only)
};
enum { CONT_DIV = 1, CONT_NOFILTER = 2, CONT_CSG = 4, CONT_XF_SHIFT = 4 };
#define CONT_XF(f) (((f) >> CONT_XF_SHIFT) & 3)
struct ChunkCounts { int32_t spheres, planes, cubes, containers; };   // per chunk, for the event counters

// per ShapeIndex (host side only, while compiling)
struct ShapeRec {
    int32_t type;        // OPK_SPHERE/PLANE/CUBE
    int32_t material;
    int32_t path_begin;  // into path_ops[]: op indices root -> shape (world_to_object chain)
    int32_t path_len;
};

// per OP INDEX of a shape op: everything shading needs to start from a hit (one 32-byte record, so
// the hit's transform chain, material and refractive index are one load away from the op index)
struct HitRec {
    int32_t shape;       // ShapeIndex (what rtx_render reports as the hit id)
    int32_t material;
    int32_t path_begin;  // into path_ops[]: op indices of the ANCESTORS, root first (the shape's own op is
    int32_t path_len;    //   not listed: it is the record's index); path_len = number of ancestors
    double ri;           // material refractive index (world.rs:258-296 reads it per container)
    int32_t type;        // OPK_SPHERE/PLANE/CUBE
    int32_t pad;
};

struct MaterialRec {     // 208 bytes
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double a[3], b[3];
    double pinv[12];     // rows 0..2 of inverse(pattern.transform)
    double pw33;
    int32_t kind;
    int32_t pdiv;        // pw33 != 1.0
};

struct LightRec { double pos[3]; double intensity[3]; };

struct SceneDev {
    const OpBody* bodies;
    const ListChunk* chunks;  // chunk 0 = first chunk of the world's list
    const ChunkCounts* chunk_counts;
    const float* list_cull;   // quads of bounding spheres in list order: {cxA,cxB,cyA,cyB,czA,czB,r2A,r2B} x 2 per 4 children
    const float* quad_cull;   // per chunk: the bounding spheres of its eight quads of children, 2 x 16 floats in the pair layout
    const float* list_cull_tight;   // list_cull / quad_cull with the tighter spheres of CSG roots (rtx_api.hip: extent_sphere); the
    const float* quad_cull_tight;   //   kernel sweeps these unless reference-faithful node-visit counts are asked for
    const int32_t* list_ops;  // op index per child slot
    const ContRec* cont_recs;
    const uint32_t* csg_prog; // the programs of all CSG roots (CSGP_* records below), walked from LDS by the kernel
    const HitRec* hits;       // n_ops + 1 entries, valid at shape ops
    const int32_t* path_ops;
    const MaterialRec* materials;
    const LightRec* lights;
    int32_t n_ops, n_shapes, n_lights, n_levels;   // n_levels = deepest list level + 1
    int32_t csg_prog_max;                          // longest CSG root program in dwords (LDS area of the interpreter)
    int32_t n_org_slots, n_dir_slots;              // LDS ray stack: origin / direction slots needed (rtx_kernel.hip)
    int32_t has_csg, has_transparency;             // has_csg: 0 none, 1 bounded (lists + program in LDS), 2 general (arena)
    int32_t csg_lds_entries, csg_prog_lds;         // LDS list entries per lane / program dwords (both 0 unless has_csg == 1)
    // has_csg == 2: the arena — arena_words * 32 slots of arena_slot_bytes, one claimed per wave for its lifetime through the
    // bitmap (a set bit = slot in use).  A slot: t[csg_entries][64] doubles, shape[csg_entries][64] u32, marks[csg_levels][2][64] u32
    unsigned char* arena; unsigned* arena_bitmap;
    int32_t arena_words, arena_slot_bytes;
    int32_t csg_entries, has_branching;  // has_branching: some material is reflective AND transparent (two children per hit)  // largest number of list entries a CSG root can produce (LDS list area per lane)
};

struct CameraDev {
    int32_t hsize, vsize;
    double pixel_size, half_width, half_height;
    double inv[12], w33;      // rows of inverse(camera.transform)
    double origin[3];         // inverse * Point(0,0,0), identical for every pixel (camera.rs:66)
    int32_t div, pad;
};

struct RenderParams {
    int32_t max_depth;
    int32_t band_first, band_stride, n_bands_local;   // bands this launch renders
    int32_t tiles_x;
    int32_t compact;          // write bands back-to-back
    int32_t no_early_out;
    // expensive-tiles-first order (rtx_api.hip): a PROBE launch traces the centre pixel of every tile at depth
    // budget 2 and appends the tile to one of RTX_TILE_CLASSES lists by the rays that took; the frame launch
    // maps block b to the b-th tile of the concatenation, most expensive class first.
    int32_t* tile_lists;       // RTX_TILE_CLASSES x n_tiles tile indices (class k at [k * n_tiles ...])
    int32_t* tile_counts;      // RTX_TILE_CLASSES list lengths (one 32-byte line)
    int32_t super_mul;         // probe launch: block b probes the 8x8 block of tiles (b * super_mul) mod n_super
    int32_t split_cap;         // > 0: up to this many tiles of the expensive classes are rendered by several waves (SPLIT kernels)
    int32_t probe, use_lists;  // probe: this IS the probe launch (no image, no counters); use_lists: follow the lists
    int32_t split_launch;     // this IS the SPLIT launch (the expensive tiles, several waves per tile)
    int32_t split_min_class;  // SPLIT kernels: tiles of this class and above are split (up to split_cap tiles)
    int32_t split_guard;      // SPLIT kernels: wave iterations after which a wave gives up (a hung wave hangs the GPU); from max_depth and n_lights
    unsigned long long* sticky;   // device-visible host word: set non-zero when this launch reports an overflow (asynchronous renders)
    int32_t tile_mul;         // block b renders tile (b * tile_mul) mod n_tiles: coprime with n_tiles, spreads heavy image
                              //   regions over the whole dispatch instead of leaving them to the tail
    // tile-cost feedback (rtx_api.hip): every frame launch leaves, per tile, the traced rays of the tile's busiest pixel (the
    // probe launch: its estimate of that figure); the next frame of the same view files its tiles by these costs with a
    // classify launch instead of a probe launch
    int32_t* tile_cost;       // n_tiles entries, or NULL
    int32_t hctx;             // hit context in LDS (rtx_kernel.hip, template flag HCTX): -1 = where the LDS budget allows, 0 / 1 = forced
};

// counters[]: 0 radiance rays, 1 shadow rays, 2 overflow pixels, 3.. events
#define RTX_CNT_RADIANCE 0
#define RTX_CNT_SHADOW 1
#define RTX_CNT_OVERFLOW 2
#define RTX_CNT_EV0 3
#define RTX_CNT_SKIPPED (3 + 24)   /* shadow rays counted but not traced (light certainly behind the surface) */
#define RTX_CNT_TOTAL (3 + 24 + 1)
// Word RTX_CNT_COST_PTR of the FIRST counter slot (a spare word: a slot has RTX_CNT_STRIDE = 32 words, RTX_CNT_TOTAL are counters)
// carries the address of the tile-cost array of this render, or 0: the frame kernels read it when a wave ends.  (As a kernel
// argument the pointer would be live — two scalar registers — through the whole kernel, also for scenes that never use it:
// measured +2 % on the CSG workload.)  The reset launch that zeroes the block writes it.
#define RTX_CNT_COST_PTR 31
#define RTX_COST_HEADER 4            /* int32 words in front of a tile-cost array: [0] = the tiles it has room for (the writer's bound) */
// The counters are replicated over RTX_CNT_SLOTS slots of RTX_CNT_STRIDE entries (256 bytes, own cache lines);
// a wave adds into slot (block index mod RTX_CNT_SLOTS) and the host sums the slots.  With ONE set of
// counters every wave's two ray-count atomics hit the same address and serialise at ~12 ns each: a floor of
// 3.1 ms for the 129 600 waves of a 4K frame (measured; it bounded S2, S3 and S5).
#ifndef RTX_TILE_CLASSES
#define RTX_TILE_CLASSES 8
#define RTX_SPLIT_MAX_TILES 20000   /* launches of at most this many tiles split their most expensive tiles over several waves */
#endif
#define RTX_ARENA_SLOTS 4096          /* general CSG mode: list slots (>= waves that can be resident: 256 CUs x 4 SIMDs x 3) */
#define RTX_MAX_LDS_BYTES 65536       /* LDS one wave (= one workgroup) may ask for */
// cost classes of the tile-order lists: class k holds the tiles whose busiest pixel traces [RTX_CLASS_LO(k), RTX_CLASS_LO(k+1)) rays
#define RTX_CLASS_OF_COST(c) ((c) < 4 ? 0 : (c) < 8 ? 1 : (c) < 16 ? 2 : (c) < 32 ? 3 : (c) < 48 ? 4 : (c) < 64 ? 5 : (c) < 96 ? 6 : 7)
#define RTX_COST_OF_CLASS(k) ((k) == 0 ? 2 : (k) == 1 ? 6 : (k) == 2 ? 12 : (k) == 3 ? 24 : (k) == 4 ? 40 : (k) == 5 ? 56 : (k) == 6 ? 80 : 120)
#define RTX_TIME_LPT_MIN_TILES 60000  /* uniform scenes: launches of at least this many tiles are ordered by measured wave durations */
#define RTX_CNT_SLOTS 256
#define RTX_GUARD_UNIT (1ull << 32)   /* the overflow counter's high half counts waves that tripped the SPLIT guard */
#define RTX_CNT_STRIDE 32

#endif
