"""Host-side launch planning (no GPU): the decode GEMV shapes (csrc/kh_model_step.hip::pick_shape) and the
prefill GEMM shapes (csrc/kh_model_prefill.hip::pg_shape) through the host-only C-ABI entries
kh_plan_decode_shapes / kh_plan_prefill_shape.  Invariants every plan must satisfy for every BASELINE
geometry, plus the plans DESIGN.md documents (so that a change of the heuristics shows up here and not
only as a slower GPU run)."""
import itertools

import pytest

from kuiperllama_amd import _ffi, binfmt

PRESETS = ["llama3.2-1b", "llama2-7b", "llama2-7b-int8", "qwen2.5-0.5b", "tinyllama-1.1b"]


@pytest.mark.parametrize("name", PRESETS)
def test_decode_plan_invariants(name):
    sp = binfmt.PRESETS[name]
    plan = _ffi.plan_decode_shapes(sp.dim, sp.hidden_dim, sp.kv_dim, sp.vocab_size, sp.quant)
    rows = {"qkv": sp.dim + 2 * sp.kv_dim, "wo": sp.dim, "ffn13": 2 * sp.hidden_dim, "w2": sp.dim,
            "cls": sp.vocab_size}
    cols = {"qkv": sp.dim, "wo": sp.dim, "ffn13": sp.dim, "w2": sp.hidden_dim, "cls": sp.dim}
    max_split = {"qkv": 2, "wo": 4, "ffn13": 1, "w2": 4, "cls": 1}
    for k, s in plan.items():
        assert s["split"] in (1, 2, 4) and s["split"] <= max_split[k], (k, s)
        # int8 w2 alone may take two exact tiles of three loads per row (round 6, profiles/r6_w2_u3_ab.txt)
        assert s["u"] in (((2, 3, 4) if k == "w2" else (2, 4)) if sp.quant else (2, 4, 8)), (k, s)
        assert s["wg"] in (256, 512), (k, s)
        assert 1 <= s["grid"] <= 1024, (k, s)  # never more than 4 x 256-thread workgroups per CU
        # a split part still streams a useful number of bytes per row pair
        elem = 1 if sp.quant else 4
        assert s["split"] == 1 or 2 * cols[k] * elem // s["split"] >= (4096 if sp.quant else 8192), (k, s)
        # the grid never exceeds the work: one workgroup handles >= one row pair per wave group
        pairs_per_wg = (s["wg"] // 64) // s["split"]
        pairs = (rows[k] // 2 + 1) // 2 if k == "ffn13" else (rows[k] + 1) // 2
        assert s["grid"] <= -(-max(pairs, 1) // pairs_per_wg) or s["grid"] % 256 == 0, (k, s)


def test_decode_plan_int8_wide_rows_keep_the_small_tile():
    """int8 rows that need more than four 16-byte loads per lane walk several tiles of TWO loads (qkv / wo / ffn13 /
    cls; w2 takes two tiles of three where that is exact) - never the fp32 rule's 4 or 8.  Geometries wider than any
    preset, so that the branch is exercised (a dangling `else` once sent every int8 launch but w2 through the fp32
    rule: invisible at the presets, whose int8 rows need at most four loads per lane)."""
    for dim, hidden in ((8192, 28672), (6144, 16384), (5120, 13824)):
        plan = _ffi.plan_decode_shapes(dim, hidden, dim, 32000, True)
        for k, s in plan.items():
            per_lane = -(-(-(-(hidden if k == "w2" else dim) // 16 // s["split"])) // 64)
            if per_lane > 4:
                want = (3,) if k == "w2" and per_lane <= 6 else (2,)
                assert s["u"] in want, (dim, hidden, k, s, per_lane)
            assert s["u"] in (2, 3, 4), (k, s)
    # fp32 keeps its own rule: eight loads per lane when the row covers them
    assert _ffi.plan_decode_shapes(8192, 28672, 8192, 32000, False)["ffn13"]["u"] == 8


def test_decode_plan_documented_shapes(monkeypatch):
    """The shapes DESIGN.md 3.2 quotes for the two bench workloads."""
    for k in ("QKV", "WO", "FFN", "W2", "CLS"):
        monkeypatch.delenv("KH_SHAPE_" + k, raising=False)
    sp = binfmt.PRESETS["llama2-7b-int8"]
    plan = _ffi.plan_decode_shapes(sp.dim, sp.hidden_dim, sp.kv_dim, sp.vocab_size, True)
    assert plan["qkv"] == {"split": 1, "u": 4, "grid": 512, "wg": 256}
    assert plan["wo"] == {"split": 1, "u": 4, "grid": 512, "wg": 256}
    assert plan["ffn13"] == {"split": 1, "u": 4, "grid": 512, "wg": 256}
    assert plan["w2"] == {"split": 2, "u": 3, "grid": 512, "wg": 512}  # 5.4 loads per lane: two tiles of three
    assert plan["cls"] == {"split": 1, "u": 4, "grid": 512, "wg": 256}
    sp = binfmt.PRESETS["llama3.2-1b"]
    plan = _ffi.plan_decode_shapes(sp.dim, sp.hidden_dim, sp.kv_dim, sp.vocab_size, False)
    assert plan["ffn13"] == {"split": 1, "u": 8, "grid": 512, "wg": 256}  # the roofline kernel of bench.py
    assert plan["w2"]["wg"] == 512 and plan["cls"]["wg"] == 512
    # the tuning hook is honoured, and a malformed value is ignored
    monkeypatch.setenv("KH_SHAPE_FFN", "1,4,256")
    assert _ffi.plan_decode_shapes(sp.dim, sp.hidden_dim, sp.kv_dim, sp.vocab_size, False)["ffn13"] == \
        {"split": 1, "u": 4, "grid": 256, "wg": 256}
    monkeypatch.setenv("KH_SHAPE_FFN", "3,4,256")
    assert _ffi.plan_decode_shapes(sp.dim, sp.hidden_dim, sp.kv_dim, sp.vocab_size, False)["ffn13"] == plan["ffn13"]


@pytest.mark.parametrize("name", PRESETS)
def test_prefill_plan_invariants(name):
    sp = binfmt.PRESETS[name]
    kq = 64 if sp.quant else 16
    gemms = [("qkv", sp.dim + 2 * sp.kv_dim, sp.dim, sp.dim % 32 == 0 and sp.kv_dim % 32 == 0),
             ("resid", sp.dim, sp.dim, sp.dim % 32 == 0), ("resid", sp.dim, sp.hidden_dim, sp.dim % 32 == 0),
             ("swiglu", sp.hidden_dim, sp.dim, sp.hidden_dim % 32 == 0)]
    for (epi, rows, K, r2), T in itertools.product(gemms, (1, 16, 17, 64, 100, 128, 129, 256, 300, 384, 512)):
        p = _ffi.plan_prefill_shape(epi, T, rows, K, sp.quant, r2)
        nm = 2 if epi == "swiglu" else 1
        assert (p["R"], p["NT"]) in ((2, 8), (2, 4), (2, 2), (1, 4)), (epi, T, p)
        assert p["R"] == 1 or r2
        assert p["slices"] * p["NT"] * 16 >= T and (p["slices"] - 1) * p["NT"] * 16 < T, (epi, T, p)
        assert p["slices"] * p["NT"] * 16 <= -(-T // 128) * 128  # inside the slab stride of the pass
        assert p["ks"] in (1, 2, 4, 8) and p["ks"] * nm * 64 <= 512, (epi, T, p)
        assert p["kz"] in (1, 2, 4) and (p["kz"] == 1 or epi == "resid"), (epi, T, p)
        # every wave keeps a useful K range (fp32: 16 blocks of 16 columns; int8: 4 blocks of 64)
        assert p["ks"] * p["kz"] == 1 or (K // kq) // (p["ks"] * p["kz"]) >= (4 if sp.quant else 16), (epi, T, p)
        assert p["workgroups"] == rows // (16 * p["R"]) * p["slices"] * p["kz"]
        # one workgroup per CU at a time: fp32 launches of more than 256 workgroups only
        assert not p["solo"] or (not sp.quant and p["workgroups"] > 256), (epi, T, p)


def test_prefill_plan_documented_shapes(monkeypatch):
    """Llama-3.2-1B, DESIGN.md 3.5: a 512-token pass puts every GEMM on the big tile; a 128-token pass splits
    the K range of wo / w2 across two workgroups; KH_PG_KZ=0 / KH_PG_SOLO=0 are read at load time and are not
    exercised here."""
    sp = binfmt.PRESETS["llama3.2-1b"]
    f = lambda epi, T, rows, K: _ffi.plan_prefill_shape(epi, T, rows, K, False, True)  # noqa: E731
    assert f("swiglu", 128, sp.hidden_dim, sp.dim) == dict(R=2, NT=8, ks=2, slices=1, solo=0, kz=1, workgroups=256)
    assert f("swiglu", 512, sp.hidden_dim, sp.dim) == dict(R=2, NT=8, ks=2, slices=4, solo=1, kz=1, workgroups=1024)
    for K in (sp.dim, sp.hidden_dim):
        assert f("resid", 512, sp.dim, K) == dict(R=2, NT=8, ks=4, slices=4, solo=0, kz=1, workgroups=256)
        p = f("resid", 128, sp.dim, K)
        assert p["kz"] == 2 and p["workgroups"] == 256 and p["solo"] == 0
    assert f("qkv", 512, sp.dim + 2 * sp.kv_dim, sp.dim)["solo"] == 1
    # Llama-2-7B fp32, 128 tokens: four-wave workgroups, one per CU at a time (was 344 two-wave workgroups)
    s7 = binfmt.PRESETS["llama2-7b"]
    p = f("swiglu", 128, s7.hidden_dim, s7.dim)
    assert (p["R"], p["NT"], p["ks"], p["solo"]) == (2, 8, 2, 1)
    # argument errors
    with pytest.raises(_ffi.KhError):
        _ffi.plan_prefill_shape("qkv", 513, 3072, 2048, False, True)
    with pytest.raises(_ffi.KhError):
        _ffi.plan_prefill_shape("qkv", 128, 3072, 2040, False, True)


def test_attention_plan_and_split_geometry(monkeypatch):
    """The decode-attention geometry (kh_attn.h::attn_plan / attn_split_len / attn_active_splits - host AND device
    code) through the host-only kh_plan_attention: the Llama-3.2-1B, Qwen2.5-0.5B, Llama-2-7B and tiny-head plans, the
    per-head / group switch, and the split arithmetic at EVERY position of a long cache against its definition
    (split length = ceil(timesteps / splits) rounded up to 64, at least 256; active = ceil(timesteps / length)) -
    the device takes a division-free shortcut for timesteps <= 256 * splits that must give the same values."""
    from kuiperllama_amd import _ffi
    monkeypatch.delenv("KH_ATTN_TLONG", raising=False)
    a = _ffi.plan_attention(32, 4, 64, 131072, 63)          # Llama-3.2-1B
    assert (a["ns"], a["ns_g"], a["stride"], a["t_long"]) == (16, 32, 32, 4096)
    assert (a["group_path"], a["active_splits"], a["split_len"], a["workgroups"]) == (0, 1, 256, 32)
    assert _ffi.plan_attention(32, 4, 64, 131072, 4094)["workgroups"] == 32 * 16
    b = _ffi.plan_attention(32, 4, 64, 131072, 4095)        # first position of the group path
    assert (b["group_path"], b["active_splits"], b["workgroups"]) == (1, 16, 8 * 16)
    assert _ffi.plan_attention(32, 4, 64, 131072, 131071)["workgroups"] == 8 * 32
    q = _ffi.plan_attention(14, 7, 64, 32768, 32767)        # Qwen2.5-0.5B: 2 KV heads cannot fill the chip with groups
    assert q["ns_g"] == 0 and q["group_path"] == 0 and q["active_splits"] == 16 and q["split_len"] == 2048
    assert _ffi.plan_attention(32, 1, 128, 4096, 4095)["ns_g"] == 0   # MHA: no group path
    assert _ffi.plan_attention(6, 1, 32, 256, 255)["ns"] == 1          # head size <= 32: generic kernel, no splits
    monkeypatch.setenv("KH_ATTN_TLONG", "300")               # the hook moves the switch point
    assert _ffi.plan_attention(32, 4, 64, 3000, 299)["group_path"] == 1
    assert _ffi.plan_attention(32, 4, 64, 3000, 298)["group_path"] == 0
    monkeypatch.delenv("KH_ATTN_TLONG")
    for seq, heads, kvm in ((9000, 8, 1), (131072, 32, 4)):
        ns = _ffi.plan_attention(heads, kvm, 64, seq, 0)["ns"]
        step = 1 if seq < 20000 else 37
        for pos in list(range(0, seq, step)) + [seq - 1]:
            p = _ffi.plan_attention(heads, kvm, 64, seq, pos)
            n = p["ns_g"] if p["group_path"] else ns
            nt = pos + 1
            want_len = max(256, (-(-nt // n) + 63) // 64 * 64)
            assert p["split_len"] == want_len and p["active_splits"] == -(-nt // want_len), (seq, pos, p)
            assert p["active_splits"] <= n
    # head size 128 (Llama-2-7B): one split up to 256 timesteps, then a 128-timestep quantum (a split then streams the
    # 128 KiB that 256 timesteps of a 64-wide head are); KH_ATTN_TS overrides the quantum
    assert _ffi.plan_attention(32, 1, 128, 2048, 0)["ns"] == 16
    for pos, (n_want, len_want) in {63: (1, 256), 255: (1, 256), 256: (3, 128), 383: (3, 128), 384: (4, 128),
                                    1023: (8, 128), 2047: (16, 128)}.items():
        p = _ffi.plan_attention(32, 1, 128, 2048, pos)
        assert (p["active_splits"], p["split_len"], p["workgroups"]) == (n_want, len_want, 32 * n_want), (pos, p)
    ns = _ffi.plan_attention(32, 1, 128, 9000, 0)["ns"]
    for pos in range(0, 9000):
        p = _ffi.plan_attention(32, 1, 128, 9000, pos)
        nt = pos + 1
        want_len = 256 if nt <= 256 else max(128, (-(-nt // ns) + 63) // 64 * 64)
        assert p["split_len"] == want_len and p["active_splits"] == -(-nt // want_len) <= ns, (pos, p)
    monkeypatch.setenv("KH_ATTN_TS", "256")
    assert _ffi.plan_attention(32, 1, 128, 2048, 383)["active_splits"] == 2
    monkeypatch.setenv("KH_ATTN_TS", "64")
    assert _ffi.plan_attention(32, 4, 64, 131072, 383)["active_splits"] == 6
    assert _ffi.plan_attention(32, 4, 64, 131072, 255)["active_splits"] == 1
    monkeypatch.delenv("KH_ATTN_TS")

