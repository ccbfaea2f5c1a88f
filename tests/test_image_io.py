"""Image output and progressive-state checkpoints (host/pthost.h).
The reference only blits dev_drawRes into a GL texture (BasicScene.cpp:424-432) and holds no
image fixtures besides screenshots, so these are format-conformance and round-trip tests:
the PNG is decoded here with zlib, the PFM and checkpoint are re-read byte for byte, and the
GPU test checks that a resumed pt_app run equals an uninterrupted one bit for bit."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import gpu_pathtracer_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "g.p.u-pathtracer_amd", "host", "pt_app")


def decode_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data) & 0xffffffff
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(chunks[1][1])
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_png_ppm_pfm_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    W, H = 301, 167                                 # > 65535 raw bytes: several stored deflate blocks
    words = rng.integers(0, 1 << 24, (H, W), dtype=np.uint32)
    rgb = np.stack([words & 255, (words >> 8) & 255, (words >> 16) & 255], axis=-1).astype(np.uint8)
    g.write_image(str(tmp_path / "a.png"), words)
    assert np.array_equal(decode_png(tmp_path / "a.png"), rgb[::-1])       # file is top row first
    g.write_image(str(tmp_path / "a.ppm"), words)
    b = open(tmp_path / "a.ppm", "rb").read()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert b.startswith(head) and np.array_equal(np.frombuffer(b[len(head):], np.uint8).reshape(H, W, 3), rgb[::-1])
    acc = rng.random((H, W, 3), dtype=np.float32) * 3 - 1
    g.write_image(str(tmp_path / "a.pfm"), acc)
    b = open(tmp_path / "a.pfm", "rb").read()
    head = f"PF\n{W} {H}\n-1.0\n".encode()
    assert b.startswith(head) and np.array_equal(np.frombuffer(b[len(head):], "<f4").reshape(H, W, 3), acc)
    with pytest.raises(RuntimeError):
        g.write_image(str(tmp_path / "nodir" / "a.png"), words)


def test_checkpoint_round_trip_and_damage(tmp_path):
    rng = np.random.default_rng(6)
    acc = rng.random((48, 80, 3), dtype=np.float32)
    path = str(tmp_path / "s.ckpt")
    g.save_checkpoint(path, acc, next_frame=37, constant_pdf=37, scene_tag=0xabcdef0123456789)
    a, nf, cp, tag = g.load_checkpoint(path)
    assert np.array_equal(a, acc) and (nf, cp, tag) == (37, 37, 0xabcdef0123456789)
    assert not os.path.exists(path + ".tmp")
    b = bytearray(open(path, "rb").read())
    b[-5] ^= 0x40                                      # one flipped bit in the pixels
    open(path, "wb").write(bytes(b))
    with pytest.raises(RuntimeError):
        g.load_checkpoint(path)
    open(path, "wb").write(bytes(b[:100]))             # truncated
    with pytest.raises(RuntimeError):
        g.load_checkpoint(path)


def read_pfm(path):
    b = open(path, "rb").read()
    parts = b.split(b"\n", 3)
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], "<f4").reshape(h, w, 3)


@pytest.mark.gpu
def test_app_resume_equals_uninterrupted_run(tmp_path):
    """pt_app: 12 samples straight == 5 samples, checkpoint, new process, 7 more (different spp per call)."""
    mesh = os.path.join(ROOT, "assets", "cornell_box.ptmesh")
    common = [APP, "--mesh", mesh, "--width", "320", "--height", "240", "--depth", "5", "--no-spheres", "--bk", "0", "0", "0"]
    run = lambda extra: subprocess.run(common + extra, check=True, capture_output=True, text=True, timeout=300).stdout
    out = run(["--frames", "12", "--spp", "4", "--out", str(tmp_path / "full.pfm")])
    assert "8 materials" in out
    run(["--frames", "5", "--spp", "5", "--checkpoint", str(tmp_path / "s.ckpt")])
    out = run(["--frames", "7", "--spp", "3", "--resume", str(tmp_path / "s.ckpt"), "--out", str(tmp_path / "resumed.pfm"),
               "--checkpoint", str(tmp_path / "s2.ckpt")])
    assert "resumed" in out
    full, resumed = read_pfm(tmp_path / "full.pfm"), read_pfm(tmp_path / "resumed.pfm")
    assert np.array_equal(full, resumed) and full.mean() > 0.005
    a, nf, cp, _ = g.load_checkpoint(str(tmp_path / "s2.ckpt"))
    assert (nf, cp) == (12, 12) and np.array_equal(a, full)
    # convert-only run: --frames 0 --resume → PNG of the checkpointed accumulator
    run(["--frames", "0", "--resume", str(tmp_path / "s2.ckpt"), "--out", str(tmp_path / "img.png")])
    img = decode_png(tmp_path / "img.png")
    expect = (np.clip(full, 0, 1) * np.float32(255)).astype(np.uint8)[::-1]
    assert np.array_equal(img, expect)
    # a checkpoint of another configuration is refused
    r = subprocess.run(common[:5] + ["--width", "160", "--height", "120", "--frames", "1", "--resume", str(tmp_path / "s.ckpt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "checkpoint belongs to another" in r.stderr


@pytest.mark.gpu
def test_app_device_build_gives_the_same_picture(tmp_path):
    """pt_app --device-build (pt_build_bvh) == pt_app over the host SAH/SBVH tree, bit for bit (the
    closest hit does not depend on the tree; postponed nothing, exact records); --fix-estimators runs."""
    mesh = os.path.join(ROOT, "assets", "bunny_low.ptmesh")
    common = [APP, "--mesh", mesh, "--width", "480", "--height", "270", "--frames", "6", "--spp", "3", "--mat", "1"]
    run = lambda extra: subprocess.run(common + extra, check=True, capture_output=True, text=True, timeout=300).stdout
    run(["--out", str(tmp_path / "host.pfm")])
    out = run(["--device-build", "--out", str(tmp_path / "dev.pfm")])
    assert "built on the device" in out
    a, b = read_pfm(tmp_path / "host.pfm"), read_pfm(tmp_path / "dev.pfm")
    assert int(np.any(a != b, axis=-1).sum()) <= 2          # wide walk: grazing candidates (test_gpu_wide.py)
    run(["--device-build", "--fix-estimators", "--out", str(tmp_path / "fix.pfm")])
    c = read_pfm(tmp_path / "fix.pfm")
    assert np.isfinite(c).all() and np.any(c != b)


@pytest.mark.gpu
def test_app_tile_split_over_contexts_is_bit_identical(tmp_path):
    """pt_app --gpus N --tile R (one context per GPU, here all on the one device; stripes gathered on the host for
    the PFM, the PNG and the checkpoint) == the unsplit run, bit for bit: SURVEY.md §8e from a C++ host."""
    mesh = os.path.join(ROOT, "assets", "gto_sixteen.ptmesh")
    common = [APP, "--mesh", mesh, "--width", "401", "--height", "233", "--frames", "5", "--spp", "2"]
    run = lambda extra: subprocess.run(common + extra, check=True, capture_output=True, text=True, timeout=300).stdout
    run(["--out", str(tmp_path / "one.pfm")])
    out = run(["--gpus", "3", "--tile", "16", "--out", str(tmp_path / "three.pfm"), "--checkpoint", str(tmp_path / "three.ckpt")])
    assert "tile split over 3 context(s)" in out
    one, three = read_pfm(tmp_path / "one.pfm"), read_pfm(tmp_path / "three.pfm")
    assert np.array_equal(one, three) and one.mean() > 0.005
    acc, nf, cp, _ = g.load_checkpoint(str(tmp_path / "three.ckpt"))
    assert (nf, cp) == (5, 5) and np.array_equal(acc, one)
    run(["--gpus", "2", "--out", str(tmp_path / "two.png")])
    run(["--out", str(tmp_path / "one.png")])
    assert np.array_equal(decode_png(tmp_path / "two.png"), decode_png(tmp_path / "one.png"))


@pytest.mark.gpu
def test_app_nee_renders_the_lit_box_with_less_noise(tmp_path):
    """pt_app --nee on CornellBox-Original (the light is the .mtl's emissive quad): at 4 spp the picture is several times
    closer to the converged one (64 spp) than 4 spp of finding the light by chance — compared below the display clamp."""
    mesh = os.path.join(ROOT, "assets", "cornell_box.ptmesh")
    common = [APP, "--mesh", mesh, "--width", "160", "--height", "120", "--depth", "4", "--no-spheres", "--bk", "0", "0", "0"]
    run = lambda extra: subprocess.run(common + extra, check=True, capture_output=True, text=True, timeout=300).stdout
    run(["--nee", "--frames", "64", "--spp", "16", "--out", str(tmp_path / "nee.pfm")])
    run(["--nee", "--frames", "4", "--spp", "4", "--out", str(tmp_path / "nee4.pfm")])
    run(["--fix-estimators", "--frames", "4", "--spp", "4", "--out", str(tmp_path / "plain4.pfm")])
    ref, nee4, plain4 = read_pfm(tmp_path / "nee.pfm"), read_pfm(tmp_path / "nee4.pfm"), read_pfm(tmp_path / "plain4.pfm")
    dim = ref.max(axis=-1) < 0.5
    assert dim.mean() > 0.4 and ref[dim].mean() > 0.02
    err_nee, err_plain = np.abs(nee4 - ref)[dim].mean(), np.abs(plain4 - ref)[dim].mean()
    print(f"4 spp against the 64-spp NEE picture: mean abs error {err_nee:.4f} with --nee, {err_plain:.4f} without")
    assert err_nee < 0.5 * err_plain
