"""The pin of the oracle: tests/golden/spirv_pixels.npz holds what the REFERENCE'S OWN compiled shaders
(shaders/*.spv, loaded by main.cpp:541-543) produce when executed by oracle/spirv_vm.py (generator:
tests/golden/make_spirv_goldens.py).  The C oracle -- an independent restatement written from the GLSL -- must
reproduce every texel and every trace count bit for bit; on the GPU box the HIP path must do the same
(tests/test_gpu_parity.py::test_spirv_*)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "spirv_pixels.npz"))
REF_SHADERS = "/root/reference/shaders/"


def _progressive(orc, scene, w, h, x, y, n_frames):
    film, rays = np.zeros((1, 1, 3), np.float32), []
    out = []
    for frame in range(n_frames):
        col, r = orc.render_rect(scene, orc.default_params(width=w, height=h, frame=frame), x, y, 1, 1, mode=1, nthreads=1)
        orc.accumulate_f32(film, col, frame)
        out.append(film.reshape(3).copy())
        rays.append(r)
    return np.array(out), rays


def test_oracle_equals_reference_shaders_progressive_1080p(orc, cornell_oracle):
    """A: 275 pixels of the 1920x1080 launch (corners, row/column 0 with their degenerate seeds, the emitter, a
    jittered grid), frames 0..2 blended as raygen.rgen:88-90 does."""
    w, h = [int(v) for v in G["a_launch"]]
    for k, (x, y) in enumerate(G["a_pixels"]):
        film, rays = _progressive(orc, cornell_oracle, w, h, int(x), int(y), G["a_texels"].shape[0])
        assert film.tobytes() == np.ascontiguousarray(G["a_texels"][:, k, :3]).tobytes(), (x, y)
        assert rays == list(G["a_traces"][:, k]), (x, y)
    assert (G["a_texels"][..., 3] == 1.0).all()  # alpha: (1 + 1*frame)/(frame+1) is exactly 1


def test_oracle_equals_reference_shaders_rgba8_image(orc, cornell_oracle):
    """B: the same launch through the reference's 8-bit storage image (imageLoad of unorm8, blend, imageStore),
    frames 0..3; fixture component order r,g,b,a, oracle memory order b,g,r,a (B8G8R8A8Unorm, main.cpp:483)."""
    w, h = [int(v) for v in G["a_launch"]]
    for k, (x, y) in enumerate(G["b_pixels"]):
        bgra = np.zeros((1, 4), np.uint8)
        for frame in range(G["b_rgba8"].shape[0]):
            col, _ = orc.render_rect(cornell_oracle, orc.default_params(width=w, height=h, frame=frame), int(x), int(y), 1, 1,
                                     nthreads=1)
            orc.accumulate_bgra8(bgra, col, frame)
            assert list(bgra[0, [2, 1, 0, 3]]) == list(G["b_rgba8"][frame, k]), (x, y, frame)


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_equals_reference_shaders_full_small_launch(orc, cornell_oracle, mode):
    """C: every invocation of a complete 120x68 launch, frames 0 and 1 (oracle brute force and LBVH walk)."""
    w, h = [int(v) for v in G["c_launch"]]
    film = np.zeros((h, w, 3), np.float32)
    for frame in (0, 1):
        col, rays, _, _ = cornell_oracle.render_frame(orc.default_params(width=w, height=h, frame=frame), mode=mode)
        orc.accumulate_f32(film, col, frame)
        assert film.tobytes() == np.ascontiguousarray(G["c_texels"][frame, :, :, :3]).tobytes()
        assert rays == int(G["c_traces"][frame].sum())


def test_oracle_golden_crop_equals_reference_shaders(orc):
    """D: the committed ORACLE golden of BASELINE config 2 (c2_crop_1080p_32spp_d8.npz, 96x64 pixels, frames 0-1)
    against the reference shaders' output for the same rectangle."""
    g = np.load(os.path.join(HERE, "golden", "c2_crop_1080p_32spp_d8.npz"))
    assert list(g["rect"]) == list(G["d_rect"])
    assert g["frame0"].tobytes() == np.ascontiguousarray(G["d_texels"][0, :, :, :3]).tobytes()
    film = g["frame0"].copy()
    orc.accumulate_f32(film, np.ascontiguousarray(g["frame1"]), 1)
    assert film.tobytes() == np.ascontiguousarray(G["d_texels"][1, :, :, :3]).tobytes()
    assert [int(r) for r in g["rays"]] == [int(r) for r in G["d_traces"].sum(axis=(1, 2))]


def _ref1024():
    path = os.path.join(HERE, "golden", "spirv_ref1024.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/spirv_ref1024.npz not generated (tests/golden/make_spirv_goldens.py --ref1024)")
    return np.load(path)


def test_oracle_equals_reference_shaders_at_the_references_own_launch(orc, cornell_oracle):
    """The reference's OWN dispatch -- traceRaysKHR(1024, 1024, 1) (main.cpp:16-17, 659), push constant frame = 0 .. 3 (main.cpp:656-658), the image blended
    in place (raygen.rgen:88-90) -- as its compiled shaders produce it (tests/golden/spirv_ref1024.npz, oracle/spirv_vm.py): a 64 x 48 rectangle through the
    tall box's front and a scattered pixel set, every texel and every trace count after every frame; a sixth of the set through the rgba8 storage image."""
    g = _ref1024()
    w, h = [int(v) for v in g["launch"]]
    assert (w, h) == (orc.default_params().width, orc.default_params().height) == (1024, 1024)
    x0, y0, rw, rh = [int(v) for v in g["rect"]]
    film = np.zeros((rh, rw, 3), np.float32)
    for frame in range(g["rect_texels"].shape[0]):
        col, rays = orc.render_rect(cornell_oracle, orc.default_params(frame=frame), x0, y0, rw, rh, mode=1, nthreads=4)
        orc.accumulate_f32(film, col, frame)
        assert film.tobytes() == np.ascontiguousarray(g["rect_texels"][frame, :, :, :3]).tobytes(), frame
        assert rays == int(g["rect_traces"][frame].sum()), frame
    for k, (x, y) in enumerate(g["pixels"]):
        f1, rays = _progressive(orc, cornell_oracle, w, h, int(x), int(y), g["texels"].shape[0])
        assert f1.tobytes() == np.ascontiguousarray(g["texels"][:, k, :3]).tobytes(), (x, y)
        assert rays == list(g["traces"][:, k]), (x, y)
    for k, (x, y) in enumerate(g["pixels_rgba8"]):
        bgra = np.zeros((1, 4), np.uint8)
        for frame in range(g["rgba8"].shape[0]):
            col, _ = orc.render_rect(cornell_oracle, orc.default_params(frame=frame), int(x), int(y), 1, 1, nthreads=1)
            orc.accumulate_bgra8(bgra, col, frame)
            assert list(bgra[0, [2, 1, 0, 3]]) == list(g["rgba8"][frame, k]), (x, y, frame)


@pytest.mark.skipif(not os.path.exists(REF_SHADERS + "raygen.rgen.spv"), reason="needs /root/reference (build container only)")
def test_fixture_is_reproducible_from_the_reference_binaries(orc, cornell_oracle):
    """Re-executes the reference's SPIR-V here for a few pixels: the committed fixture is what the binaries give,
    also with the driver's closest hit taken from the LBVH walk instead of brute force."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_spirv_goldens as gen
    from oracle import spirv_vm as vm
    w, h = [int(v) for v in G["a_launch"]]
    gen._init()
    for k in (0, 3, 57, 140, 201, 274):
        x, y = [int(v) for v in G["a_pixels"][k]]
        tex, rays = gen.run_pixel((x, y, w, h, 2, False))
        assert np.array(tex, np.float32).tobytes() == np.ascontiguousarray(G["a_texels"][:2, k]).tobytes()
        assert rays == list(G["a_traces"][:2, k])
    m = vm.Module(REF_SHADERS + "raygen.rgen.spv")
    assert m.version == 0x00010600 and "main" in m.names.values()  # SPIR-V 1.6 (shaders/compile.bat)


def test_spirv_vm_refuses_what_it_does_not_know(tmp_path):
    """the interpreter raises on any opcode outside the set the three shaders use, never guesses"""
    import struct
    from oracle import spirv_vm as vm
    p = tmp_path / "x.spv"
    # header + OpTypeVoid %1 + OpTypeFunction %2 %1 + OpFunction %1 %3 None %2 + OpLabel %4 + OpKill + OpFunctionEnd
    words = [0x07230203, 0x00010600, 0, 5, 0, (2 << 16) | 19, 1, (3 << 16) | 33, 2, 1, (5 << 16) | 54, 1, 3, 0, 2,
             (2 << 16) | 248, 4, (1 << 16) | 252, (1 << 16) | 56]
    p.write_bytes(struct.pack("<%dI" % len(words), *words))
    m = vm.Module(str(p))
    pipe = vm.Pipeline.__new__(vm.Pipeline)
    pipe.drv, pipe.n_instructions, pipe.n_traces = None, 0, 0
    with pytest.raises(NotImplementedError):
        pipe._call(m, m.functions[3], [], list(m.const))
    (tmp_path / "bad.spv").write_bytes(b"\0" * 32)
    with pytest.raises(ValueError):
        vm.Module(str(tmp_path / "bad.spv"))
