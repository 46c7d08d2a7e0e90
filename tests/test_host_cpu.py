"""CPU-side tests of the product's host code (no GPU, no compute calls on the device):
the shared library loads and exports every symbol include/*.h declares, the C++ host mirror
computes the same bits as the oracle for transforms / inverses / bounds, and the C++ YAML
loader agrees byte for byte with the independent Python loader."""
import ctypes
import math
import os
import re
import subprocess

import numpy as np
import pytest

import ora
import rtx
import scenes
import yaml_scene
from rtapi import CUBE, PLANE, SPHERE, UNION, csg, group, shape

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = set()
    for h in ("rtx.h", "rtx_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"^\s*(?:int|void|int64_t|const char\*|rtxh_world\*)\s+(rtxh?_[a-z_0-9]+)\s*\(", text, flags=re.M))
    assert declared == set(rtx.RTX_SYMBOLS + rtx.RTXH_SYMBOLS), declared ^ set(rtx.RTX_SYMBOLS + rtx.RTXH_SYMBOLS)
    for name in declared:
        assert getattr(rtx.LIB, name) is not None


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rtx.RtxError) as e:
        rtx.Context(0)
    assert e.value.code == rtx.E_DEVICE


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "ray-tracer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".py", "Makefile")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "rt_oracle" not in text and "import ora" not in text, os.path.join(dp, f)


def _same(a, b):
    a, b = np.asarray(list(a), np.float64), np.asarray(list(b), np.float64)
    if not np.array_equal(a, b, equal_nan=True):
        return False
    ok = ~np.isnan(a)                      # NaN payload/sign is not observable by the renderer
    return np.array_equal(np.signbit(a[ok]), np.signbit(b[ok]))


def test_transforms_bit_equal_to_oracle():
    rng = np.random.default_rng(7)
    P, O = rtx.API, ora.API
    for _ in range(200):
        a = float(rng.uniform(-7, 7))
        for ax in "xyz":
            assert _same(getattr(P, "rotation_" + ax)(a), getattr(O, "rotation_" + ax)(a))
        ops = []
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(0, 5))
            if k == 0:
                ops.append(("scale", [float(x) for x in rng.uniform(0.1, 3, 3)]))
            elif k == 1:
                ops.append(("translate", [float(x) for x in rng.uniform(-5, 5, 3)]))
            else:
                ops.append((["rotate-x", "rotate-y", "rotate-z"][k - 2], float(rng.uniform(-4, 4))))
        tp, to = P.compose(ops), O.compose(ops)
        assert _same(tp, to)
        assert _same(P.inverse(tp), O.inverse(to))
        f, t, u = rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 3)
        assert _same(P.view_transform(f, t, u), O.view_transform(f, t, u))
    cp, co = P.camera(400, 200, math.pi / 3), O.camera(400, 200, math.pi / 3)
    assert bytes(cp) == bytes(co)
    assert P.rotation_x(0.0)[:] == P.identity()[:]


def _build_tree(api):
    w = api.world_new()
    g1 = w.add_group(api.compose([("rotate-y", 0.7), ("translate", [1, 2, 3])]))
    g2 = w.add_group(api.compose([("scale", [1, 2, 3]), ("rotate-z", -0.4)]))
    s1 = w.add_shape(SPHERE, None, api.compose([("scale", [0.5, 0.5, 0.5]), ("rotate-x", 1.1), ("translate", [5, 0, 0])]))
    s2 = w.add_shape(CUBE, None, api.compose([("rotate-z", 0.3), ("translate", [0, -2, 0.5])]))
    p = w.add_shape(PLANE, None, api.identity())
    gp = w.add_group(api.identity())
    c = w.add_csg(UNION, shape(w.add_shape(SPHERE, None, api.translation(0.5, 0, 0))),
                  shape(w.add_shape(CUBE, None, api.rotation_y(0.9))), api.scaling(2, 1, 1))
    w.add_object_to_scene(group(g1)); w.set_group_of(group(g2), g1)
    w.set_group_of(shape(s1), g2); w.set_group_of(shape(s2), g2); w.set_group_of(csg(c), g1)
    w.set_group_of(shape(p), gp); w.add_object_to_scene(group(gp))
    w.build()
    return w, [group(g1), group(g2), group(gp), csg(c)]


def test_bounds_bit_equal_to_oracle_including_the_from_points_quirk():
    wp, objs = _build_tree(rtx.API)
    wo, _ = _build_tree(ora.API)
    for o in objs:
        assert _same(wp.get_bounds(o), wo.get_bounds(o)), o
    # a group holding only a plane keeps NaN bounds (inf * 0 corners)
    assert all(math.isnan(v) for v in wp.get_bounds(objs[2]))


@pytest.mark.parametrize("name,kw", [("S1", {}), ("S2", {}), ("S3", dict(levels=2, per_leaf=3)), ("S4", {}), ("S5", {})])
def test_cpp_yaml_loader_equals_python_loader(name, kw):
    text = scenes.make(name, **kw)
    w1, c1 = rtx.load_yaml(text)
    w2, c2 = yaml_scene.load(text, rtx.API)
    assert rtx.serialize(w1) == rtx.serialize(w2)
    assert len(c1) == len(c2) == 1 and bytes(c1[0]) == bytes(c2[0])


def test_yaml_expression_quirks():
    # yaml.rs:466-509: "a op b"; the operator is never cleared, so "1 + 2 3" = 6 and "1 + 2 * 3" is NaN
    def fov(expr):
        _, cams = rtx.load_yaml("- add: camera\n  width: 10\n  height: 10\n  field-of-view: %s\n" % expr)
        hv = cams[0].half_width
        return 2 * math.atan(hv)
    assert abs(fov("pi / 2") - math.pi / 2) < 1e-15
    assert abs(fov("1. / pi") - 1 / math.pi) < 1e-15          # yaml.rs:539-543
    assert abs(fov("0.25 + 0.25 0.5") - 1.0) < 1e-15
    assert math.isnan(fov("1 + 2 * 3"))
    assert abs(fov("1") - 1.0) < 1e-15 and abs(fov("0x1") - 1.0) < 1e-15
    _, cams = rtx.load_yaml("- add: camera\n  width: 160\n  height: 120\n  field-of-view: pi / 2\n")   # yaml.rs:517-530
    assert list(cams[0].transform) == list(rtx.API.identity())


def test_yaml_errors_instead_of_panics():
    for bad in ["- define: group\n  transform:\n    - translate: [0, 1, 0]\n",         # Group without name (yaml.rs:331)
                "- add: sphere\n  material: nope\n",                                    # unknown material
                "- add: csg\n  operator: union\n",                                      # missing operands (yaml.rs:213)
                "- add: csg\n  operator: xor\n  operands:\n    - add: sphere\n    - add: cube\n",
                "- add: camera\n  width: 1.5\n  height: 10\n  field-of-view: 1\n",      # width must be Integer (yaml.rs:77)
                "just a string\n"]:
        with pytest.raises(rtx.RtxError):
            rtx.load_yaml(bad)
    w, cams = rtx.load_yaml("- add: frobnicator\n- define: thing\n- add: sphere\n  colour: red\n")   # warnings only
    assert cams == [] and len(rtx.serialize(w)) > 0


def test_yaml_flow_and_comments():
    text = ("# comment\n- add: light   # trailing\n  at: [ -10 , 10, -10 ]\n  intensity: [1, 1, 1]\n"
            "- define: material\n  name: 'my mat'\n  color: [1, 0.5, 0]\n"
            "- {add: sphere, material: my mat}\n"
            "- add: cube\n  transform:\n  - scale: [2, 2, 2]\n  - rotate-y: -pi / 4\n")
    w1, _ = rtx.load_yaml(text)
    w2, _ = yaml_scene.load(text, rtx.API)
    assert rtx.serialize(w1) == rtx.serialize(w2)
    # quoted scalars: escapes and doubled quotes are unescaped (the reader keeps views of the text for everything else),
    # a quoted number is a string, a '#' inside quotes is no comment, a repeated key keeps its last value
    text = ("- define: material\n  name: \"a \\\"b\\\" #1\"\n  color: [0.25, 0.5, 1]\n"
            "- define: material\n  name: 'it''s'\n  extends: \"a \\\"b\\\" #1\"\n  ambient: 0.75\n  ambient: 0.5\n"
            "- define: material\n  name: twelve\n  extends: 'it''s'\n  diffuse: 1e-1\n"
            "- add: sphere\n  material: \"twelve\"\n"
            "- {add: cube, material: 'it''s', transform: [{translate: [1, +2, -.5e1]}]}\n")
    w1, _ = rtx.load_yaml(text)
    w2, _ = yaml_scene.load(text, rtx.API)
    assert rtx.serialize(w1) == rtx.serialize(w2)
    # a quoted number is a String (yaml-rust types plain scalars only; the PyYAML-based restatement cannot tell them apart)
    quoted = "- define: material\n  name: '12'\n  ambient: 0.5\n- add: sphere\n  material: \"12\"\n"
    plain = "- define: material\n  name: m\n  ambient: 0.5\n- add: sphere\n  material: m\n"
    assert rtx.serialize(rtx.load_yaml(quoted)[0]) == rtx.serialize(rtx.load_yaml(plain)[0])
    with pytest.raises(rtx.RtxError):
        rtx.load_yaml("- add: sphere\n  material: 12\n")          # an unquoted 12 is an Integer: as_str().unwrap() (yaml.rs:120)


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    rtx.write_png(p, img)
    back = np.asarray(Image.open(p))
    assert back.shape == img.shape and np.array_equal(back, img)


def test_png_writer_large_image_is_one_valid_stream(tmp_path):
    """write_png deflates row blocks on several threads and concatenates them (sync-flushed blocks + one final block, zlib
    header and Adler-32 around them): a frame large enough for many blocks must decode to the same pixels with an
    independent decoder, including ragged block boundaries."""
    import time
    from PIL import Image
    rng = np.random.default_rng(2)
    h, w = 1031, 2049
    img = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] // np.array([3, 5, 7])).astype(np.uint8)   # smooth, like a render
    img[::17, ::13] = rng.integers(0, 256, img[::17, ::13].shape, dtype=np.uint8)
    p = str(tmp_path / "big.png")
    t0 = time.perf_counter()
    rtx.write_png(p, img)
    dt = time.perf_counter() - t0
    Image.MAX_IMAGE_PIXELS = None
    back = np.asarray(Image.open(p))
    assert back.shape == img.shape and np.array_equal(back, img)
    print("write_png %dx%d: %.0f ms, %.1f MB" % (w, h, dt * 1e3, os.path.getsize(p) / 1e6))


def test_cli_usage_message():
    out = subprocess.run([os.path.join(ROOT, "ray-tracer_amd", "rtx")], capture_output=True, text=True)
    assert "Need to provide filename of scene description as argument" in out.stdout     # main.rs:16


def test_scene_desc_validation_needs_no_gpu():
    # rtx_scene_create rejects malformed input before touching the device; here it cannot even get a ctx
    assert rtx.LIB.rtx_scene_create(None, None, None) == rtx.E_INVALID
    assert rtx.LIB.rtx_band_rows(100, rtx.Bands(0, 1)) == 104
    assert rtx.LIB.rtx_band_rows(100, rtx.Bands(1, 2)) == 48 and rtx.LIB.rtx_band_rows(100, rtx.Bands(0, 2)) == 56
    assert rtx.LIB.rtx_strerror(rtx.E_CAPACITY) is not None


# ---------------------------------------------------------------- loader fuzz
def _fz_num(rng, ints=True):
    k = int(rng.integers(0, 8))
    if k == 0 and ints:
        return str(int(rng.integers(-5, 6)))
    if k == 1:
        return "%.17g" % float(rng.uniform(-5, 5))
    if k == 2:
        return ["pi / 2", "pi / 3.", "-pi / 4", "pi * 0.25", "1. / pi", "2 * pi", "pi - 1", "0.5 + 0.25"][int(rng.integers(0, 8))]
    if k == 3:
        return ["1e-1", "2.5E+0", ".5", "3.", "+2", "-0.0", "0x2", "1 + 2 3"][int(rng.integers(0, 8))]
    return "%.6g" % float(rng.uniform(0.1, 3))


def _fz_tuple(rng):
    sp = ["", " "][int(rng.integers(0, 2))]
    return "[" + ("," + sp).join(_fz_num(rng) for _ in range(3)) + "]"


def _fz_transform(rng, ind):
    lines = []
    for _ in range(int(rng.integers(1, 5))):
        k = ["scale", "translate", "rotate-x", "rotate-y", "rotate-z", "shear"][int(rng.integers(0, 6))]
        v = _fz_tuple(rng) if k in ("scale", "translate", "shear") else _fz_num(rng)
        lines.append("%s- %s: %s" % (ind, k, v))
    return lines


def _fz_material_keys(rng, ind, names):
    keys = []
    for k in rng.permutation(["color", "pattern", "ambient", "diffuse", "specular", "shininess", "reflective",
                              "transparency", "refractive-index", "extends", "colour"]):
        if rng.random() < 0.45:
            continue
        if k == "color":
            keys.append(["%scolor: %s" % (ind, _fz_tuple(rng))])
        elif k == "pattern":
            ty = ["stripes", "rings", "checkers", "gradient", "uniform", "plain", "plaid"][int(rng.integers(0, 7))]
            p = ["%spattern:" % ind, "%s  type: %s" % (ind, ty)]
            if ty in ("uniform", "plain"):
                p.append("%s  color: %s" % (ind, _fz_tuple(rng)))
            else:
                p += ["%s  color1: %s" % (ind, _fz_tuple(rng)), "%s  color2: %s" % (ind, _fz_tuple(rng))]
            if rng.random() < 0.6:
                p += ["%s  transform:" % ind] + _fz_transform(rng, ind + "    ")
            keys.append(p)
        elif k == "extends":
            if names:
                keys.append(["%sextends: %s" % (ind, names[int(rng.integers(0, len(names)))])])
        elif k == "colour":
            keys.append(["%scolour: red" % ind])                     # unknown key: warning only
        else:
            keys.append(["%s%s: %s" % (ind, k, _fz_num(rng))])
    return keys


def _fz_shape(rng, ind, mats, groups, operand):
    kind = ["sphere", "cube", "plane"][int(rng.integers(0, 3))]
    lines = ["add: " + kind]
    for k in rng.permutation(["material", "transform", "group", "shadow"]):
        if k == "material" and mats and rng.random() < 0.7:
            lines.append("material: " + (mats[int(rng.integers(0, len(mats)))] if rng.random() > 0.02 else "nope"))
        elif k == "transform" and rng.random() < 0.7:
            lines += ["transform:"] + _fz_transform(rng, "  ")
        elif k == "group" and groups and not operand and rng.random() < 0.5:
            lines.append("group: " + (groups[int(rng.integers(0, len(groups)))] if rng.random() > 0.02 else "nope"))
        elif k == "shadow" and rng.random() < 0.1:
            lines.append("shadow: false")
    return [ind + l for l in lines]


def _fz_csg(rng, ind, mats, depth):
    lines = ["add: csg"]
    for k in rng.permutation(["operator", "transform", "operands"]):
        if k == "operator":
            lines.append("operator: " + ["union", "intersection", "difference"][int(rng.integers(0, 3))])
        elif k == "transform" and rng.random() < 0.5:
            lines += ["transform:"] + _fz_transform(rng, "  ")
        elif k == "operands":
            lines.append("operands:")
            for _ in range(2):
                sub = (_fz_csg(rng, "", mats, depth + 1) if depth < 2 and rng.random() < 0.3
                       else _fz_shape(rng, "", mats, [], True))
                lines += ["  - " + sub[0]] + ["    " + l for l in sub[1:]]
    return [ind + l for l in lines]


def _fz_doc(rng):
    out, mats, groups = [], [], []

    def item(lines):
        out.append("- " + lines[0]); out.extend("  " + l for l in lines[1:])

    for _ in range(int(rng.integers(3, 14))):
        k = int(rng.integers(0, 8))
        if k == 0:
            cam = ["add: camera", "width: %d" % rng.integers(1, 400), "height: %d" % rng.integers(1, 300),
                   "field-of-view: " + _fz_num(rng, ints=False)]
            if rng.random() < 0.8:
                cam += ["from: " + _fz_tuple(rng), "to: " + _fz_tuple(rng), "up: " + _fz_tuple(rng)]
            item(cam)
        elif k == 1:
            item(["add: light", "at: " + _fz_tuple(rng), "intensity: " + _fz_tuple(rng)])
        elif k == 2:
            name = "m%d" % len(mats)
            body = _fz_material_keys(rng, "", mats)
            body.insert(int(rng.integers(0, len(body) + 1)), ["name: " + name])   # 'name' anywhere among the keys
            item(["define: material"] + [l for k in body for l in k])
            mats.append(name)
        elif k == 3:
            name = "g%d" % len(groups)
            g = ["define: group", "name: " + name]
            if groups and rng.random() < 0.6:
                g.append("parent: " + groups[int(rng.integers(0, len(groups)))])
            if rng.random() < 0.7:
                g += ["transform:"] + _fz_transform(rng, "  ")
            item(g); groups.append(name)
        elif k == 4:
            item(_fz_csg(rng, "", mats, 0))
        elif k == 5 and rng.random() < 0.3:
            item(["add: teapot", "size: 3"])                          # unknown 'add': warning only
        else:
            item(_fz_shape(rng, "", mats, groups, False))
        if rng.random() < 0.15:
            out.append("# comment %d" % len(out))
    return "\n".join(out) + "\n"


def test_yaml_loader_fuzz_against_python_loader():
    """N2: the C++ loader and the independent PyYAML-based restatement must issue the same builder calls
    (serialised world + cameras byte-identical) on generated scene files; both must reject the same files."""
    rng = np.random.default_rng(20261111)
    n_ok = n_cam = 0
    for i in range(300):
        text = _fz_doc(rng)
        try:
            w2, c2 = yaml_scene.load(text, rtx.API)
        except (yaml_scene.SceneError, KeyError):
            with pytest.raises(rtx.RtxError):
                rtx.load_yaml(text)
            continue
        w1, c1 = rtx.load_yaml(text)
        assert rtx.serialize(w1) == rtx.serialize(w2), "doc %d:\n%s" % (i, text)
        assert len(c1) == len(c2), text
        for a, b in zip(c1, c2):
            # NaN fields (e.g. a degenerate view transform) compare as bytes too: same operations, same NaNs
            assert _same(np.frombuffer(bytes(a), np.float64, offset=8), np.frombuffer(bytes(b), np.float64, offset=8)), text
            assert bytes(a)[:8] == bytes(b)[:8]
        n_ok += 1; n_cam += len(c1)
    assert 240 < n_ok < 300 and n_cam > 50


def _rust_structs(path):
    """#[repr(C)] structs of integration/rust/src/gpu.rs -> {name: [(field, rust type)]}"""
    import re
    src = open(path).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\n\}", src, re.S):
        fields = re.findall(r"pub (\w+):\s*([^,\n]+),", m.group(2))
        out[m.group(1)] = [(f, t.strip()) for f, t in fields]
    return out


def _repr_c_layout(fields):
    """size, {field: offset} under Rust's repr(C) rules (= the C ABI's) on x86-64"""
    import re
    prim = {"i32": (4, 4), "u32": (4, 4), "f64": (8, 8), "u64": (8, 8), "i64": (8, 8), "u8": (1, 1)}

    def size_align(t):
        if t.startswith("*"):
            return 8, 8
        m = re.match(r"\[(\w+);\s*(\d+)\]", t)
        if m:
            s, a = prim[m.group(1)]
            return s * int(m.group(2)), a
        return prim[t]
    off, offs, amax = 0, {}, 1
    for name, t in fields:
        s, a = size_align(t)
        off = (off + a - 1) // a * a
        offs[name] = off
        off += s
        amax = max(amax, a)
    return (off + amax - 1) // amax * amax, offs


def test_rust_binding_layouts_match_the_c_header(tmp_path):
    """N1: the Rust shim in integration/rust cannot be compiled here (no rustc), but its #[repr(C)] structs
    can be pinned: the layout Rust's repr(C) rules give each struct in gpu.rs (sizes and field offsets, computed here) must
    equal what gcc lays out for the struct of the same role in include/rtx.h (sizeof / offsetof from a compiled C
    program), field by field and in the same order; and every function of its extern "C" block must be exported by
    librtx_hip.so with the same number of parameters as the prototype in rtx.h."""
    import re
    import subprocess
    import rtx
    root = ROOT
    rs = _rust_structs(os.path.join(root, "integration", "rust", "src", "gpu.rs"))
    pairs = {"RtxMaterial": "rtx_material", "RtxSceneDesc": "rtx_scene_desc", "RtxCamera": "rtx_camera",
             "RtxBands": "rtx_bands", "RtxStats": "rtx_stats"}
    assert set(pairs) <= set(rs), sorted(rs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rtx.h"', 'int main(void) {']
    for r, c in pairs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (r, c))
        for f, _ in rs[r]:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (r, f, c, f))      # a field missing in C fails to compile
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        a, b, v = line.split()
        got[(a, b)] = int(v)
    header = open(os.path.join(root, "include", "rtx.h")).read()
    for r, c in pairs.items():
        size, offs = _repr_c_layout(rs[r])
        assert got[(r, "size")] == size, (r, got[(r, "size")], size)
        for f, o in offs.items():
            assert got[(r, f)] == o, (r, f, got[(r, f)], o)
        # same number of fields: nothing of the C struct is missing on the Rust side
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (c, c), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_c = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_c == len(rs[r]), (r, n_c, len(rs[r]))
    # the extern "C" block against the header's prototypes and the library's exports
    gpu_rs = open(os.path.join(root, "integration", "rust", "src", "gpu.rs")).read()
    block = gpu_rs[gpu_rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    fns = re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, re.S)
    assert len(fns) >= 15
    flat = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, args in fns:
        getattr(rtx.LIB, name)                                                 # exported
        proto = re.search(r"\b%s\s*\((.*?)\);" % name, flat, re.S)
        assert proto, name
        c_args = [a for a in proto.group(1).split(",") if a.strip() and a.strip() != "void"]
        r_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, c_args, r_args)
