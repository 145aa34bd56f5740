"""The C++ host layer (include/rtx_dropin.hpp: parse, create_mesh, create_boxs,
init_scene, flatten) against the CPU oracle's scene assembly and the structure fixtures
that come from the reference's own functions.  Bit for bit."""
import hashlib
import importlib
import json
import os
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("ray-tracing_amd.host_py")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("cfg", ["default", "spheres", "bunny", "dino_spheres", "buddha", "multi"])
def test_host_scene_equals_reference_structure(op, scenes, host, golden_dir, cfg):
    with open(os.path.join(golden_dir, "structure.json")) as f:
        want = json.load(f)[cfg]
    hb = host.HostBuilder(10)
    d = scenes.assemble(hb, cfg)
    assert (d.n_nodes, d.n_leaves, d.n_prims, d.n_materials) == (
        want["n_nodes"], want["n_leaves"], want["n_prims"], want["n_materials"])
    for k, v in op.desc_arrays(d).items():
        assert sha(v) == want["sha256_" + k], k


@pytest.mark.parametrize("depth", [0, 2, 7, 12])
def test_host_scene_equals_oracle_at_other_depths(op, scenes, host, depth):
    hb = host.HostBuilder(depth)
    a = op.desc_arrays(scenes.assemble(hb, "dino_spheres"))
    ob = op.OracleBuilder(depth)
    b = op.desc_arrays(scenes.assemble(ob, "dino_spheres"))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _write_off(path, verts, faces, header="OFF", counts=None):
    with open(path, "w") as f:
        f.write("%s\n%s\n" % (header, counts or "%d %d 0" % (len(verts), len(faces))))
        for v in verts:
            f.write("%.9g %.9g %.9g\n" % tuple(float(x) for x in v))
        for t in faces:
            f.write("3 %d %d %d \n" % tuple(int(x) for x in t))


def test_parse_round_trip_and_errors(op, scenes, host, tmp_path):
    verts, faces = scenes.load_mesh("bunny")
    path = str(tmp_path / "bunny.off")
    _write_off(path, verts, faces)
    rc, v, f = host.parse_off(path)
    assert rc == 0
    assert np.array_equal(v.view(np.uint32), verts.view(np.uint32)) and np.array_equal(f, faces)
    bad = str(tmp_path / "bad.off")
    _write_off(bad, verts[:3], faces[:0], header="OFF ")
    assert host.parse_off(bad)[0] == 1                       # meshs.h:79-83
    assert host.parse_off(str(tmp_path / "nope.off"))[0] == 1  # unopened stream reads as "not OFF"


def test_parse_keeps_the_reference_face_count_quirk(op, host, tmp_path):
    """meshs.h:88-91 reads the face count from (digits of the vertex count + 1)
    characters: '8 12 0' yields 1 face, in the host layer as in the oracle."""
    verts = np.arange(24, dtype=np.float32).reshape(8, 3)
    faces = np.array([[k % 8, (k + 1) % 8, (k + 2) % 8] for k in range(12)], np.uint32)
    path = str(tmp_path / "cube.off")
    _write_off(path, verts, faces)
    rc_h, v_h, f_h = host.parse_off(path)
    rc_o, v_o, f_o = op.parse_off(path)
    assert rc_h == rc_o == 0
    assert len(f_h) == len(f_o) == 1 and np.array_equal(f_h, f_o) and np.array_equal(v_h, v_o)


def test_init_scene_from_files(op, scenes, host, golden_dir, tmp_path):
    """init_scene() (scene.h:4-108) reading meshs/*.off from the working directory."""
    os.makedirs(tmp_path / "meshs")
    for name in ("bunny", "dino"):
        v, f = scenes.load_mesh(name)
        _write_off(str(tmp_path / "meshs" / (name + ".off")), v, f)
    hb = host.HostBuilder(10)
    hb.init_scene(str(tmp_path))
    d = hb.build()
    with open(os.path.join(golden_dir, "structure.json")) as f:
        want = json.load(f)["init_scene"]
    for k, v in op.desc_arrays(d).items():
        assert sha(v) == want["sha256_" + k], k


def test_init_scene_fails_without_meshes(host, tmp_path):
    hb = host.HostBuilder(10)
    with pytest.raises(RuntimeError):
        hb.init_scene(str(tmp_path))


def test_png_writer(host, tmp_path):
    """8-bit RGB PNG; row j = 0 is the bottom scanline (FreeImage's convention)."""
    rgb = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    host.write_png(path, rgb)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert ihdr == (53, 37, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 3)
    assert (raw[:, 0] == 0).all()
    assert np.array_equal(raw[:, 1:].reshape(37, 53, 3), rgb[::-1])


@pytest.mark.skipif(not os.path.exists("/root/reference/main.cpp"), reason="reference tree not present")
def test_reference_main_builds_against_the_drop_in(tmp_path):
    """The reference's main.cpp with its one include line switched to the drop-in header
    (the one edit a maintainer makes) compiles as C++11 and links against the two
    libraries; nothing else in it changes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open("/root/reference/main.cpp").read()
    assert text.count('#include "raytracing.h"') == 1
    (tmp_path / "main_dropin.cpp").write_text(text.replace('#include "raytracing.h"', '#include "rtx_dropin.hpp"'))
    cmd = ["g++", "-std=c++11", "-I", os.path.join(root, "include"), str(tmp_path / "main_dropin.cpp"),
           "-o", str(tmp_path / "main"),
           "-L", os.path.join(root, "ray-tracing_amd", "host"), "-lrtx_host",
           "-L", os.path.join(root, "ray-tracing_amd", "csrc"), "-lrtx_amd"]
    subprocess.check_call(cmd)


def _mesh_dir(scenes, where):
    os.makedirs(os.path.join(where, "meshs"))
    for name in ("bunny", "dino"):
        v, f = scenes.load_mesh(name)
        _write_off(os.path.join(where, "meshs", name + ".off"), v, f)


RT_MAIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray-tracing_amd", "host",
                       "rt_main")


def test_rt_main_prints_what_main_cpp_prints(scenes, rtx, tmp_path):
    """The command-line renderer's text follows main.cpp:9-31: banner, start time and the
    first progress line before the frame.  Without a GPU the render step then fails loudly
    (exit code 1, message on stderr, no 'Finish time'); the complete text is checked by the
    GPU test below."""
    import re
    import subprocess
    _mesh_dir(scenes, str(tmp_path))
    out = subprocess.run([RT_MAIN, "--width", "64", "--height", "48", "--lights", "2"], cwd=str(tmp_path),
                         capture_output=True, timeout=300)
    head = r"RAY TRACING by Romain Roy\n-------------------------\nStart time: \d+:\d+:\d+\nRendering image\.\.\. 0 %\r"
    assert re.match(head, out.stdout.decode()), out.stdout
    if rtx.device_count() == 0:
        assert out.returncode == 1 and b"render_image:" in out.stderr and b"Finish time" not in out.stdout
    # a scene file that is not an OFF file ends the program before anything is rendered
    (tmp_path / "meshs" / "dino.off").write_text("PLY\n")
    out = subprocess.run([RT_MAIN], cwd=str(tmp_path), capture_output=True, timeout=300)
    assert out.returncode == 1 and b"Start time" not in out.stdout


@pytest.mark.gpu
def test_rt_main_renders_and_reports(scenes, tmp_path):
    """main.cpp:9-31 and raytracing.h:132-141 end to end on the GPU: the whole text, and
    out.png in the working directory."""
    import re
    import subprocess
    _mesh_dir(scenes, str(tmp_path))
    out = subprocess.run([RT_MAIN, "--width", "320", "--height", "200", "--lights", "3"], cwd=str(tmp_path),
                         capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr
    want = (r"RAY TRACING by Romain Roy\n-------------------------\nStart time: \d+:\d+:\d+\n"
            r"Rendering image\.\.\. 0 %\r(Rendering image\.\.\. \d+ %\r){19}Rendering image\.\.\. 100 %\r\nImage successfully saved!\n"
            r"Finish time: \d+:\d+:\d+\nCompute time: 0:0:\d+\n"
            r"Compute time \(ms\): [0-9.]+ \(320 x 200, 3 light samples, 1 GPU\(s\)\)\n$")
    assert re.match(want, out.stdout.decode()), out.stdout
    assert (tmp_path / "out.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"


@pytest.mark.skipif(not os.path.exists("/root/reference/meshs/bunny.off"), reason="reference tree not present")
def test_mesh_files_hold_what_parse_reads_from_the_reference(host, scenes):
    """ray-tracing_amd/data/*.mesh are the vertices and indices that parse() (meshs.h:60-125)
    returns for the reference's own OFF files (bit for bit)."""
    for name in ("bunny", "dino", "buddha"):
        rc, v, f = host.parse_off("/root/reference/meshs/%s.off" % name)
        assert rc == 0
        v0, f0 = scenes.load_mesh(name)
        assert np.array_equal(v.view(np.uint32), v0.view(np.uint32)) and np.array_equal(f, f0)


DROPIN_SNIPPET = r'''
#include "rtx_dropin.hpp"
#include <cstdio>
#include <cstdlib>
// a caller that assembles its own values with the reference's vector operators
int main(int argc, char **argv) {
    if (argc != 4) return 2;
    const int n = std::atoi(argv[3]);
    FILE *in = std::fopen(argv[1], "rb"), *out = std::fopen(argv[2], "wb");
    if (!in || !out) return 3;
    for (int k = 0; k < n; k++) {
        Vec3F v;
        if (std::fread(&v, sizeof(v), 1, in) != 1) return 4;
        const Vec3F u = normalize(v);
        const Vec3F w = {v.z, v.x, v.y};
        const Vec3F r = reflect(u, normalize(w));
        const Vec3F c = cross(v, w);
        const Vec3F mix = (v + w) * 0.5f - (v * w) / 3.f + v / w;
        const float s[4] = {dot(v, w), dot2(v), norm(v), (v < w ? 1.f : 0.f) + (v > w ? 2.f : 0.f) +
                                                          (v == w ? 4.f : 0.f) + (v != w ? 8.f : 0.f)};
        std::fwrite(&u, sizeof(u), 1, out);
        std::fwrite(&r, sizeof(r), 1, out);
        std::fwrite(&c, sizeof(c), 1, out);
        std::fwrite(&mix, sizeof(mix), 1, out);
        std::fwrite(s, sizeof(s), 1, out);
    }
    Intersections hits;
    Intersection hit = Intersection();
    hit.distance = 1.f;
    hit.normale = Vec3F{0.f, 0.f, 1.f};
    hit.object.type = SPHERE;
    hits.push_back(hit);
    std::fclose(out);
    return hits.size() == 1 && sizeof(Intersection) == 100 ? 0 : 5; // structs.h:45-51: 72 + 12 + 4 + 12 bytes
}
'''


def test_vector_operators_of_the_drop_in_header(op, golden_dir, tmp_path):
    """vec3.h:9-112 and structs.h:45-51,74 as the drop-in header carries them: a C++11
    caller that uses the operators compiles, normalize() returns the bits the reference's
    normalize returned for the committed known-answer inputs, and the other functions
    follow the reference's evaluation order (checked against a binary32 restatement, and
    reflect also against the reference's own function where its harness is built)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kat = np.load(os.path.join(golden_dir, "kat_functions.npz"))
    v = np.ascontiguousarray(kat["normalize_in"], dtype=np.float32)
    (tmp_path / "snippet.cpp").write_text(DROPIN_SNIPPET)
    v.tofile(str(tmp_path / "in.bin"))
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-Wall", "-Wextra", "-I",
                           os.path.join(root, "include"), str(tmp_path / "snippet.cpp"), "-o", str(tmp_path / "snippet")])
    subprocess.check_call([str(tmp_path / "snippet"), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(len(v))])
    got = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float32).reshape(len(v), 16)
    f = np.float32
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    assert np.array_equal(bits(got[:, 0:3]), bits(kat["normalize_out"]))
    # binary32 restatement, operation by operation
    with np.errstate(all="ignore"):
        w = v[:, [2, 0, 1]]
        dot_ = lambda a, b: (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]
        nrm = lambda a: a / np.sqrt(dot_(a, a))[:, None]
        u, n = nrm(v), nrm(w)
        refl = u - (n * f(2.0)) * dot_(n, u)[:, None]
        cr = np.stack([v[:, 1] * w[:, 2] - v[:, 2] * w[:, 1], v[:, 2] * w[:, 0] - v[:, 0] * w[:, 2],
                       v[:, 0] * w[:, 1] - v[:, 1] * w[:, 0]], axis=1)
        mix = ((v + w) * f(0.5) - (v * w) / f(3.0)) + v / w
        lt, gt = (v < w).all(axis=1), (v > w).all(axis=1)
        eq = (v == w).all(axis=1)
        flags = lt * f(1) + gt * f(2) + eq * f(4) + (~eq) * f(8)
    assert np.array_equal(bits(got[:, 3:6]), bits(refl))
    assert np.array_equal(bits(got[:, 6:9]), bits(cr))
    assert np.array_equal(bits(got[:, 9:12]), bits(mix))
    assert np.array_equal(bits(got[:, 12]), bits(dot_(v, w)))
    assert np.array_equal(bits(got[:, 13]), bits(dot_(v, v)))
    assert np.array_equal(bits(got[:, 14]), bits(np.sqrt(dot_(v, v))))
    assert np.array_equal(got[:, 15], flags)
    if op.have_ref():   # the reference's own reflect (vec3.h:103-106)
        import ctypes as C
        ref = op.ref()
        F3 = C.c_float * 3
        for k in range(0, len(v), 7):
            o = F3()
            ref.ref_reflect(F3(*u[k]), F3(*n[k]), o)
            assert np.array_equal(bits(np.array(list(o), dtype=np.float32)), bits(got[k, 3:6]))
