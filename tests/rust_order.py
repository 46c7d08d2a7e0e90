"""The scene description integration/rust/src/flatten.rs would hand to rtx_scene_create, rebuilt in Python from a world's
canonical dump (rtxh_world_serialize): node ids = shapes first (in ShapeIndex order), then groups, then CSGs (both in
creation order); one material per shape (material id = ShapeIndex); inverses taken with the host mirror's inverse.  Used to
check that the C ABI accepts that node order and renders the same image as the C++ flattener's (creation order)."""
import ctypes as C
import struct

import numpy as np

import rtx

NODE_GROUP, NODE_CSG = 3, 4


def parse(blob):
    off = 0

    def i32():
        nonlocal off
        v = struct.unpack_from("<i", blob, off)[0]; off += 4
        return v

    def f64(n):
        nonlocal off
        v = list(struct.unpack_from("<%dd" % n, blob, off)); off += 8 * n
        return v
    nodes = []
    for _ in range(i32()):
        n = dict(kind=i32(), parent=i32(), transform=f64(16))
        n["children"] = [i32() for _ in range(i32())]
        n["csg_op"] = i32(); n["shape_index"] = i32()
        n["shader"] = f64(7); n["a"] = f64(3); n["b"] = f64(3); n["ptransform"] = f64(16); n["pkind"] = i32()
        n["bounds"] = f64(6)
        nodes.append(n)
    scene = [i32() for _ in range(i32())]
    lights = [(f64(3), f64(3)) for _ in range(i32())]
    assert off == len(blob)
    return nodes, scene, lights


def inverse(m16):
    a = (C.c_double * 16)(*m16); o = (C.c_double * 16)()
    rtx.LIB.rtxh_inverse(a, o)
    return list(o)


def desc_in_rust_order(world):
    """-> (SceneDesc, keepalive) with the node numbering of flatten.rs"""
    nodes, scene, lights = parse(rtx.serialize(world))
    shapes = sorted([k for k, n in enumerate(nodes) if n["kind"] < NODE_GROUP], key=lambda k: nodes[k]["shape_index"])
    assert [nodes[k]["shape_index"] for k in shapes] == list(range(len(shapes)))
    groups = [k for k, n in enumerate(nodes) if n["kind"] == NODE_GROUP]
    csgs = [k for k, n in enumerate(nodes) if n["kind"] == NODE_CSG]
    order = shapes + groups + csgs
    new_id = {old: new for new, old in enumerate(order)}
    n = len(order)
    kind = np.array([nodes[k]["kind"] for k in order], np.int32)
    inv = np.array([inverse(nodes[k]["transform"]) for k in order], np.float64).reshape(-1)
    bounds = np.array([nodes[k]["bounds"] for k in order], np.float64).reshape(-1)
    first, count, children = np.zeros(n, np.int32), np.zeros(n, np.int32), []
    for new, old in enumerate(order):
        first[new] = len(children)
        children += [new_id[c] for c in nodes[old]["children"]]
        count[new] = len(nodes[old]["children"])
    csg_op = np.array([nodes[k]["csg_op"] for k in order], np.int32)
    shape_index = np.array([nodes[k]["shape_index"] if nodes[k]["kind"] < NODE_GROUP else -1 for k in order], np.int32)
    material = np.array([nodes[k]["shape_index"] if nodes[k]["kind"] < NODE_GROUP else -1 for k in order], np.int32)
    mats = (rtx.RtxMaterial * max(len(shapes), 1))()
    for s, k in enumerate(shapes):
        nd = nodes[k]
        m = mats[s]
        (m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index) = nd["shader"]
        for j in range(3):
            m.color_a[j] = nd["a"][j]; m.color_b[j] = nd["b"][j]
        pinv = inverse(nd["ptransform"])
        for j in range(16):
            m.pattern_inverse[j] = pinv[j]
        m.pattern_kind = nd["pkind"]
    children = np.array(children if children else [0], np.int32)
    roots = np.array([new_id[r] for r in scene] if scene else [0], np.int32)
    lpos = np.array([l[0] for l in lights] if lights else [[0, 0, 0]], np.float64).reshape(-1)
    lint = np.array([l[1] for l in lights] if lights else [[0, 0, 0]], np.float64).reshape(-1)
    pi, pd = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    d = rtx.SceneDesc(n_nodes=n, node_kind=kind.ctypes.data_as(pi), node_inverse=inv.ctypes.data_as(pd),
                      node_bounds=bounds.ctypes.data_as(pd), node_first_child=first.ctypes.data_as(pi),
                      node_child_count=count.ctypes.data_as(pi), node_csg_op=csg_op.ctypes.data_as(pi),
                      node_shape_index=shape_index.ctypes.data_as(pi), node_material=material.ctypes.data_as(pi),
                      n_children=sum(int(c) for c in count), children=children.ctypes.data_as(pi),
                      n_roots=len(scene), roots=roots.ctypes.data_as(pi), n_materials=len(shapes), materials=mats,
                      n_lights=len(lights), light_position=lpos.ctypes.data_as(pd), light_intensity=lint.ctypes.data_as(pd))
    return d, (kind, inv, bounds, first, count, children, csg_op, shape_index, material, mats, roots, lpos, lint)
