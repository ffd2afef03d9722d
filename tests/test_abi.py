"""The C-ABI library loads without a GPU and exports every symbol include/socialways_hip.h declares;
the packed-weight layout it reports is the one the Python modules pack their state_dicts into."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(section="product"):
    """Entry points the header declares: the product surface, or the measurement section behind it."""
    src = open(os.path.join(ROOT, "include", "socialways_hip.h")).read()
    product, marker, measurement = src.partition("/* ==== MEASUREMENT SECTION")
    assert marker, "include/socialways_hip.h lost its measurement-section marker"
    src = re.sub(r"/\*.*?\*/", "", product if section == "product" else "/*" + measurement, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from socialways_amd import _lib as L
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), "include/socialways_hip.h declares %s but the library does not export it" % name
    assert sorted(L.PROTOTYPES) == syms, "socialways_amd/_lib.py binds exactly the header's entry points"
    dbg = header_symbols("measurement")
    assert dbg == sorted(L.DEBUG_PROTOTYPES) == ["sw_debug_spin", "sw_kernel_timing", "sw_kernel_timing_read"]
    assert not set(dbg) & set(syms) and all(hasattr(lib, n) for n in dbg)
    assert lib.sw_version() >= 1
    assert isinstance(lib.sw_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (no compute call is made here)."""
    from socialways_amd import _lib as L
    lib = L.load()
    assert lib.sw_traj_4d(None, None, 4, 8, 12, None, None, None) == -1
    assert lib.sw_enc_lstm_fwd(None, 0, None, None, None, 4, 8, None, None, None, None, None, 0, None) == -1
    assert lib.sw_param_count(99, 12) == -1 and lib.sw_param_offset(0, 99, 1) == -1
    assert lib.sw_gan_loss(None, None, 0, None, None, None, 0, 8, 1.0, 1.0, None, None, None, None, None, None, None) == -1
    # the data-parallel exchange (csrc/sw_comm.hip): sizes and argument checks are host-side
    assert lib.sw_comm_bytes(0, 1000) == -1 and lib.sw_comm_bytes(17, 1000) == -1 and lib.sw_comm_bytes(8, 0) == -1
    assert lib.sw_comm_bytes(8, 86124) > 2 * 8 * 86124 and lib.sw_comm_bytes(2, 86124) % 16 == 0      # two granule regions
    assert lib.sw_comm_bytes(8, 1 << 28) == -2                        # 32-bit byte offsets inside a buffer: refused, not wrapped
    assert lib.sw_allreduce_direct(None, 0, 2, 1000, None, 10, None) == -1
    assert lib.sw_allreduce_direct_adam(None, 0, 2, 1000, None, 10, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 0, None) == -1
    assert lib.sw_comm_alloc(0, None) == -1 and lib.sw_comm_free(None) == 0 and lib.sw_comm_ipc_close(None) == 0


@pytest.mark.parametrize("tp", [2, 12])
def test_packed_layout_matches_state_dicts(tp):
    """Packed buffers = state_dict tensors in key order on 4-float boundaries (SURVEY.md §2.2 shapes)."""
    import socialways_amd as sw
    from socialways_amd import _lib as L
    lib = L.load()
    torch.manual_seed(0)
    G = sw.Generator()
    D = sw.Discriminator(tp, 64, 2)
    for grp, mod, n_ref in ((L.GRP_ENC, G.encoder, 33600), (L.GRP_EMB, G.feature_embedder, 6400),
                            (L.GRP_ATT, G.attention, 4160), (L.GRP_DEC, G.decoder, 41962),
                            (L.GRP_DISC, D, 27939 if tp == 12 else 26659)):
        sd = mod.state_dict()
        assert sum(v.numel() for v in sd.values()) == n_ref
        assert lib.sw_param_tensors(grp) == len(sd)
        flat = mod._flat
        assert flat.numel() == lib.sw_param_count(grp, tp)
        end = 0
        for i, (k, v) in enumerate(sd.items()):
            off = lib.sw_param_offset(grp, i, tp)
            assert off % 4 == 0 and off >= end, (k, off)
            assert torch.equal(flat[off:off + v.numel()].view(v.shape), v), k
            assert v.data_ptr() == flat.data_ptr() + 4 * off, "%s is a view of the packed buffer" % k
            end = off + v.numel()
    # load_state_dict writes through to the packed buffer
    sd = {k: torch.randn_like(v) for k, v in D.state_dict().items()}
    D.load_state_dict(sd)
    off = lib.sw_param_offset(L.GRP_DISC, 19, tp)
    assert torch.equal(D._flat[off:off + 2], sd["latent_decoder.2.bias"])


def test_state_dict_keys_equal_reference_modules():
    import socialways_amd as sw
    import sw_oracle as O
    G, D = sw.Generator(), sw.Discriminator(12, 64, 2)
    o = O.SocialWaysOracle(12)
    for a, b in ((G.encoder, o.encoder), (G.feature_embedder, o.feature_embedder), (G.attention, o.attention),
                 (G.decoder, o.decoder), (D, o.D)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert [tuple(v.shape) for v in sa.values()] == [tuple(v.shape) for v in sb.values()]


def test_same_seed_same_initial_weights_as_reference_order():
    """Construction order encoder, feature_embedder, attention, decoder, D on the CPU generator
    (train.py:370-385): seed 0 reproduces the reference's initial weights bit for bit."""
    import socialways_amd as sw
    from _util import golden, state_from
    g = golden("toy_b64_on")
    torch.manual_seed(0)
    G = sw.Generator()
    D = sw.Discriminator(2, 64, 2)
    w0 = state_from(g, "w0.")
    for name, mod in (("encoder", G.encoder), ("feature_embedder", G.feature_embedder), ("attention", G.attention),
                      ("decoder", G.decoder), ("D", D)):
        for k, v in mod.state_dict().items():
            assert torch.equal(v, w0[name][k]), (name, k)


def test_workspace_sizes():
    from socialways_amd import _lib as L
    B, To, Tp = 2048, 8, 12
    assert L.workspace_floats(L.WS_GSAVE, B, To, Tp) == B * ((To + Tp - 1) * 388 + Tp * 280)
    assert L.workspace_floats(L.WS_GDELTA, B, To, Tp) == B * ((To + Tp - 1) * 256 + Tp * 284 + 160)
    assert L.workspace_floats(L.WS_PAIRS, B, To, Tp, 1, 16384) == B * 200 + 16384 * 196   # per-agent rows dWh | Wh | Q | sd, pair rows h2 | dh2 | h1 | dh1 | feat
    assert L.workspace_floats(L.WS_DSAVE, B, To, Tp, 2) > L.workspace_floats(L.WS_DSAVE, B, To, Tp, 1)


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors / without the library; it never computes on the host."""
    import socialways_amd as sw
    G = sw.Generator(use_social=True)
    with pytest.raises(sw.SocialWaysHipError):
        G(torch.rand(4, 8, 2), torch.rand(4, 32), 12, [[0, 4]])
    with pytest.raises(sw.SocialWaysHipError):
        sw.get_traj_4d(torch.rand(4, 8, 2), [])
    with pytest.raises(sw.SocialWaysHipError):
        sw.Discriminator(12, 64, 2)(torch.rand(4, 8, 4), torch.rand(4, 12, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "socialways_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "sw_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("tp", [2, 12, 20])
def test_disc_image_table_describes_the_documented_layouts(tp):
    """sw_disc_image_table (host function, no GPU): every packed Discriminator float maps to its places in the image
    buffer - the MFMA A-operand images of lstm.weight_hh / its transpose and the transposed, zero-padded head
    matrices; rebuilt here independently with numpy from the layout comments of csrc/sw_common.h / sw_disc.hip."""
    from socialways_amd import _lib as L
    lib = L.load()
    n = lib.sw_param_count(L.GRP_DISC, tp)
    nimg = lib.sw_disc_image_floats(tp)
    tab = np.empty((n, 2), dtype=np.int32)
    assert lib.sw_disc_image_table(tp, tab.ctypes.data) == 0
    w = np.arange(1, n + 1, dtype=np.float64)            # distinct non-zero values
    img = np.zeros(nimg)
    used = tab[tab >= 0]
    assert used.max() < nimg and len(np.unique(used)) == len(used), "image places are distinct and inside the buffer"
    for c in range(2):
        m = tab[:, c] >= 0
        img[tab[m, c]] = w[m]
    off = [lib.sw_param_offset(L.GRP_DISC, i, tp) for i in range(20)]
    whh = w[off[1]:off[1] + 256 * 64].reshape(256, 64)
    lane = np.arange(64)
    ln, lg = lane & 15, lane >> 4
    # A-operand image of M [rows][K]: float4 of (row tile t, k-step j, lane l) = M[16 t + ln][16 j + 4 lg .. + 3]
    op = img[:16384].reshape(16, 4, 64, 4)
    opT = img[16384:32768].reshape(4, 16, 64, 4)
    for t in range(16):
        for j in range(4):
            for e in range(4):
                assert np.array_equal(op[t, j, :, e], whh[16 * t + ln, 16 * j + 4 * lg + e])
    whhT = whh.T                                           # [64][256]
    for t in range(4):
        for j in range(16):
            for e in range(4):
                assert np.array_equal(opT[t, j, :, e], whhT[16 * t + ln, 16 * j + 4 * lg + e])
    # heads: XT [K (padded)][ld] row-major, in the order of0 of1 pe0 pe1 cl0 la0 (ld 36) cl1 la1 (ld 20)
    kp = (4 * tp + 15) // 16 * 16
    pos = 32768
    for idx, (M, K, rows, ld) in zip((4, 6, 8, 10, 12, 16, 14, 18),
                                     ((32, 64, 64, 36), (32, 32, 32, 36), (32, 4 * tp, kp, 36), (32, 32, 32, 36),
                                      (32, 64, 64, 36), (32, 64, 64, 36), (1, 32, 32, 20), (2, 32, 32, 20))):
        X = w[off[idx]:off[idx] + M * K].reshape(M, K)
        blk = img[pos:pos + rows * ld].reshape(rows, ld)
        want = np.zeros((rows, ld))
        want[:K, :M] = X.T
        assert np.array_equal(blk, want), idx
        pos += rows * ld
    assert pos == nimg
    assert (tab[off[0]:off[1]] < 0).all() and (tab[off[2]:off[4]] < 0).all(), "W_ih and the biases have no image"
