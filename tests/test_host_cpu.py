"""CPU-only checks of the host side: C-ABI symbols, program builder, synthetic weights, sharding,
gloo all-gather.  No HIP compute here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import build_program, synth_state_dict, zoo_meta, make_meta
from yololite_amd import dist as ydist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "yololite_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yl_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    """build() has produced the .so; it loads and exports exactly what include/yololite_hip.h declares."""
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _header_symbols()
    assert declared == sorted(n for n, _, _ in _lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().yl_abi_version() == _lib.YL_ABI_VERSION
    assert _lib.load().yl_strerror(-5).decode() == "unsupported configuration"


def test_no_cpu_fallback_fails_loudly():
    """Without a HIP device the product path raises; it never routes to a CPU implementation."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    meta = zoo_meta("edge_n", 3, 64)
    m = ya.build_model_from_meta(meta)
    m.load_state_dict(synth_state_dict(meta))
    with pytest.raises(ya.YoloLiteHipError):
        m.to("cuda:0")
    with pytest.raises(ya.YoloLiteHipError):
        m.to("cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "yololite-official-repo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_program_edge_n_macs_and_layout():
    """edge_n C=80 640^2 = 0.7964 GMAC over 63 conv layers; fused into 42 launches."""
    meta = zoo_meta("edge_n", 80, 640)
    p = build_program(meta, synth_state_dict(meta), fuse_dw=True, fuse_stem=False, fuse_uib=False)
    assert abs(p.macs - 796.39e6) < 0.05e6
    assert p.level_size == [80, 40, 20] and p.level_anchors == [1, 1, 1] and p.strides == [8, 16, 32]
    assert [p.slots[p.feature_slots[k]] for k in ("c3", "c4", "c5")] == [(80, 80, 32), (40, 40, 48), (20, 20, 480)]
    # SURVEY App. A counts the head box/obj/cls convs as one row; a fused inverted-residual launch (yl_ir_kernel: c2 > 0)
    # holds three convs (expand, depthwise, project), a dense conv with a chained 1x1 (c3 > 0) two
    nconv = sum(1 + (l.dw_k > 0) + (l.op == 1 and l.c2 > 0) + (l.op == 1 and l.c3 > 0) for l in p.layers)
    assert nconv == 63
    p2 = build_program(meta, synth_state_dict(meta), fuse_dw=False, fuse_stem=False, fuse_uib=False)
    assert len(p2.layers) > len(p.layers) and p2.macs == p.macs
    p3 = build_program(meta, synth_state_dict(meta), fuse_dw=True, fuse_stem=True, fuse_uib=False)
    # 3 entry convs -> 1 launch (without the fused entry: the stem + [blocks.0.0 with blocks.0.1 chained] = 2 launches)
    assert len(p3.layers) == len(p.layers) - 1 and p3.macs == p.macs and p3.layers[0].op == 3
    # by default (yl_ir_kernel, round 3) the four 40x40 blocks and the two lateral + smooth pairs at 80x80 / 40x40 are
    # one launch each; fuse_uib adds the four 20x20 blocks through the per-wave kernel
    p0 = build_program(meta, synth_state_dict(meta), fuse_dw=True, fuse_stem=True, fuse_uib=False, fuse_ir=False)
    assert len(p3.layers) == len(p0.layers) - 6 and p0.macs == p.macs
    p4 = build_program(meta, synth_state_dict(meta), fuse_uib=True)
    assert len(p4.layers) == len(p0.layers) - 10 and p4.macs == p.macs


@pytest.mark.parametrize("name,feat", [("edge_m", [(80, 80, 64), (40, 40, 96), (20, 20, 960)]),
                                       ("yololite_m", [(80, 80, 48), (40, 40, 120), (20, 20, 352)]),
                                       ("yololite_n_v2", [(80, 80, 48), (40, 40, 112), (20, 20, 192)]),
                                       ("yololite_m_v2", [(80, 80, 56), (40, 40, 120), (20, 20, 208)])])
def test_program_other_configs(name, feat):
    meta = zoo_meta(name, 80, 640)
    p = build_program(meta, synth_state_dict(meta))
    assert [p.slots[p.feature_slots[k]] for k in ("c3", "c4", "c5")] == feat
    assert p.level_size == [80, 40, 20]


def test_state_dict_keys_match_oracle_model():
    """Key set / shapes the builder consumes == the reference-compatible module's state_dict."""
    from oracle import model as om
    for name in ("edge_n", "yololite_m", "yololite_m_v2"):
        meta = zoo_meta(name, 7, 128, use_p6=(name == "edge_n"))
        sd = synth_state_dict(meta)
        m = om.build_from_meta(meta)
        msd = m.state_dict()
        assert set(sd) <= set(msd)
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(msd[k].shape), k
        p = build_program(meta, msd)
        assert [k for k in msd if k not in p.known_keys] == []


def test_fused_block_query_is_the_librarys_own_table():
    """program.py used to carry hand-copied mirrors of the kernels' shape tables; it now asks the library
    (yl_query_fused_block: host-side code of the .so).  Spot values + what the programs built from it look like."""
    from yololite_amd import _lib
    q = _lib.load().yl_query_fused_block
    assert q(48, 96, 48, 3, 1, 40, 40) == 1            # edge_n's 40x40 UIB blocks: workgroup-level-halo kernel
    assert q(16, 96, 24, 3, 2, 160, 160) == 1          # yololite_m blocks.1.0
    assert q(64, 256, 64, 5, 1, 20, 20) == 2           # 20x20 grids: the per-wave kernel only (fuse_uib)
    assert q(48, 96, 48, 3, 1, 41, 41) == 0 and q(48, 96, 48, 7, 1, 40, 40) == 0 and q(0, 96, 48, 3, 1, 40, 40) == 0
    meta = zoo_meta("edge_n", 80, 640)
    n_all = len(build_program(meta, synth_state_dict(meta)).layers)
    n_dw3 = len(build_program(meta, synth_state_dict(meta), fuse_dw="dw3").layers)
    assert n_dw3 > n_all                                # "dw3": the 5x5 depthwise convs stay separate launches everywhere


def test_bench_kernel_labels_follow_the_launchers_selection():
    """bench.kernel_family names the kernel the dispatcher picks for a layer (the `kernel` string of the roofline object).  The
    round-5 kernels work on 8 x 8-pixel windows / 4 x 4-tile m-tiles and take a layer only where the grid fills them
    (yl_launch_conv_wino / yl_launch_conv_dwk, yl_convc.hip): pinned here for yololite_m at the benchmark's shape so that the
    label cannot drift from the launcher unnoticed (round 5: yl_conv_dws_kernel had taken over layers whose label said
    yl_conv_dwk_kernel)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    meta = zoo_meta("yololite_m", 80, 640)
    prog = build_program(meta, synth_state_dict(meta, seed=1))
    wl = bench.wino_layers(prog, 1)
    fam = {}
    for k, L in enumerate(prog.layers):
        hw = tuple(prog.slots[L.out_slot][:2])
        fam.setdefault(bench.kernel_family(L, k in wl, hw, 32), []).append(hw[0])
    assert sorted(fam["yl_conv_wino2_kernel"]) == [20, 20, 40, 40, 80, 80]       # every FPN 3x3 conv (20 x 20 fills 69 %)
    assert sorted(fam["yl_conv_dwl_kernel"]) == [40, 40, 80, 80]                 # head trunks at 80 x 80 and 40 x 40
    assert sorted(fam["yl_conv_dwk_kernel"]) == [20, 20]                         # ... the 20 x 20 ones keep the tap-load kernel
    assert "yl_conv_wino_kernel" not in fam
    # a batch too small for three windows per CU stays on the tap-load kernel
    head80 = next(L for L in prog.layers if L.dw_k == 3 and L.cin == 328 and prog.slots[L.out_slot][0] == 80)
    assert bench.kernel_family(head80, False, (80, 80), 4) == "yl_conv_dwk_kernel"


def test_missing_weight_raises_and_meta_errors():
    meta = zoo_meta("edge_n", 3, 64)
    sd = synth_state_dict(meta)
    del sd["lateral4.bias"]
    with pytest.raises(RuntimeError):
        ya.build_model_from_meta(meta).load_state_dict(sd)
    bad = dict(meta, arch="nope")
    with pytest.raises(ValueError):
        ya.build_model_from_meta(bad)
    nokey = dict(meta, config=dict(model=meta["config"]["model"], training={}))
    with pytest.raises(KeyError):                       # tools/infer.py:49-50 behaviour
        ya.build_model_from_meta(nokey)


def test_shard_range_partitions():
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [ydist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import yololite_amd
from yololite_amd import dist as ydist
rank, world, total = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(sys.argv[2])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[3]}", rank=rank, world_size=world)
g = torch.Generator().manual_seed(5)
full_d = torch.randn(total, 4, 6, generator=g)
full_c = torch.randint(0, 5, (total,), generator=g, dtype=torch.int32)
def fake_predict(x):            # stands in for HipContext.predict on this rank's shard
    lo, hi = ydist.shard_range(total, rank, world)
    assert x.shape[0] == hi - lo
    return full_d[lo:hi].clone(), full_c[lo:hi].clone()
d, c = ydist.sharded_predict(fake_predict, torch.zeros(total, 1), "cpu")
assert torch.equal(d, full_d) and torch.equal(c, full_c), (rank, d.shape)
if total % world == 0:          # zero-copy gatherer (equal shards): results written in place, one collective
    b = total // world
    gat = ydist.DetGatherer(b, 4, "cpu")
    lo, hi = ydist.shard_range(total, rank, world)
    for it in range(3):                                  # pipelined: gather() returns the previous step's result
        gat.dets.copy_(full_d[lo:hi] + it); gat.counts.copy_(full_c[lo:hi] + it)
        prev = gat.gather()
        if it == 0:
            assert prev is None
        else:
            assert torch.equal(prev[0].reshape(total, 4, 6), full_d + (it - 1))
            assert torch.equal(prev[1].reshape(total), full_c + (it - 1))
    gd, gc = gat.flush()
    assert torch.equal(gd.reshape(total, 4, 6), full_d + 2) and torch.equal(gc.reshape(total), full_c + 2)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("total", [8, 5])
def test_allgather_dets_gloo_world2(tmp_path, total):
    """N>1 path on CPU: two gloo ranks, sharded 'predict', one all-gather, image order preserved
    (also with an uneven shard)."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = 29600 + (os.getpid() + total) % 300
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(total), str(port)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_WORKER8 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import yololite_amd
from yololite_amd import dist as ydist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
total, max_out, steps = 512, 300, 3
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=rank, world_size=world)
b = total // world
lo, hi = ydist.shard_range(total, rank, world)
assert (lo, hi) == (rank * b, (rank + 1) * b)
def shard_result(step, r):       # what rank r's yl_predict would leave in its gather slot at `step`: image-index stamped
    g = torch.Generator().manual_seed(1000 * step + r)
    d = torch.randn(b, max_out, 6, generator=g)
    d[:, 0, 0] = torch.arange(r * b, (r + 1) * b, dtype=torch.float32)
    c = torch.randint(0, max_out + 1, (b,), generator=g, dtype=torch.int32)
    return d, c
gat = ydist.DetGatherer(b, max_out, "cpu")
assert gat.world == 8
def check(views, step):
    gd, gc = views
    assert gd.shape == (world, b, max_out, 6) and gc.shape == (world, b)
    for r in range(world):
        d, c = shard_result(step, r)
        assert torch.equal(gd[r], d) and torch.equal(gc[r], c), (rank, step, r)
    flat = gd.reshape(total, max_out, 6)[:, 0, 0]            # image i of the global batch is (i // b, i % b)
    assert torch.equal(flat, torch.arange(total, dtype=torch.float32))
for step in range(steps):
    d, c = shard_result(step, rank)
    gat.dets.copy_(d); gat.counts.copy_(c)
    prev = gat.gather()
    assert (prev is None) == (step == 0)
    if prev is not None:
        check(prev, step - 1)
check(gat.flush(), steps - 1)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_config5_shape_gloo_world8(tmp_path):
    """BASELINE config 5's shape on CPU: 8 gloo ranks, B=512 -> 64 images per rank, max_out 300, the pipelined
    DetGatherer over 3 steps (one all-gather of [dets | counts] per step, result of step i returned at step i+1)."""
    script = tmp_path / "w8.py"
    script.write_text(_WORKER8)
    port = 29950 + os.getpid() % 40
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(port)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_WORKER_LANES = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import yololite_amd
from yololite_amd import dist as ydist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
B, max_out, steps, K = 64, 1024, 7, 2            # bench.py --gpus 2 --in-flight 2: B images per rank, packed rows of 1024
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=rank, world_size=world)
def shard_result(step, r):                        # what rank r's yl_predict of step `step` leaves in its lane's gather slot
    g = torch.Generator().manual_seed(1000 * step + r)
    d = torch.randn(B, max_out, 6, generator=g)
    d[:, 0, 0] = torch.arange(r * B, (r + 1) * B, dtype=torch.float32) + 10000.0 * step
    c = torch.randint(0, max_out + 1, (B,), generator=g, dtype=torch.int32)
    return d, c
gats = [ydist.DetGatherer(B, max_out, "cpu") for _ in range(K)]      # one gatherer per lane (bench.measure_predict: gats[k])
issued = []                                        # (lane, slot) of every collective in issue order
def check(views, step):
    gd, gc = views
    for r in range(world):
        d, c = shard_result(step, r)
        assert torch.equal(gd[r], d) and torch.equal(gc[r], c), (rank, step, r)
    flat = gd.reshape(world * B, max_out, 6)[:, 0, 0]                # image i of the global batch is (i // B, i % B)
    assert torch.equal(flat, torch.arange(world * B, dtype=torch.float32) + 10000.0 * step)
for step in range(steps):
    k = step % K                                   # ServingPipeline.run: step i on lane i % K
    g = gats[k]
    d, c = shard_result(step, rank)
    g.dets.copy_(d); g.counts.copy_(c)            # lane_work: yl_predict(out=(gats[k].dets, gats[k].counts))
    issued.append((k, g._k))
    prev = g.gather()                              # ... return gats[k].gather(): the lane's previous exchange
    assert (prev is None) == (step < K), (step, prev is None)
    if prev is not None:
        check(prev, step - K)
for k in range(K):                                 # bench.block(): pipe.flush(); for g in gats: g.flush()
    last = max(s for s in range(steps) if s % K == k)
    check(gats[k].flush(), last)
seqs = [None] * world
dist.all_gather_object(seqs, issued)
assert all(s == seqs[0] for s in seqs), seqs      # the collective issue order is the same on every rank
assert issued == [(s % K, (s // K) % 2) for s in range(steps)]
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_bench_shape_two_lanes_two_gatherers_gloo_world2(tmp_path):
    """the multi-GPU bench shape under --in-flight 2 on CPU -- two lanes, each with its OWN DetGatherer
    (two slots each), 7 steps, two gloo ranks with 64 images and packed rows of 1024 each.  Step i writes lane i % 2's
    current slot and starts its all-gather; what comes back is that lane's previous step, whole and in image order; the
    order in which collectives are issued is identical on every rank (it is program order: lane i % 2, slot (i // 2) % 2);
    the flush at the end of a timed block returns each lane's last step."""
    script = tmp_path / "wl.py"
    script.write_text(_WORKER_LANES)
    port = 29400 + os.getpid() % 150
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(port)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_linear_weight_stored_as_1x1_conv_is_accepted():
    """timm's ConvNeXt variants with conv_mlp=True keep mlp.fc1 / fc2 as 1x1 Conv2d ([4c, c, 1, 1]) instead of
    nn.Linear ([4c, c]); the builder takes either layout and packs the same layers."""
    meta = make_meta(img_size=64, arch="YOLOLiteMS", backbone="oracle_tiny_cnx", num_classes=3, fpn_channels=16,
                     depth_multiple=0.5, head_depth=1)
    sd = dict(synth_state_dict(meta, seed=5))
    build_program(meta, sd)                                   # materialises every key of the synthetic dict
    sd = {k: np.asarray(v) for k, v in sd.items()}
    p0 = build_program(meta, sd)
    sd4 = {k: (v.reshape(v.shape + (1, 1)) if (k.endswith("mlp.fc1.weight") or k.endswith("mlp.fc2.weight")) else v)
           for k, v in sd.items()}
    assert any(v.ndim == 4 and k.endswith("mlp.fc1.weight") for k, v in sd4.items())
    p1 = build_program(meta, sd4)
    assert len(p0.layers) == len(p1.layers)
    for a, b in zip(p0.layers, p1.layers):
        for f in ("w", "b", "w2", "b2"):
            u, v = getattr(a, f), getattr(b, f)
            assert (u is None) == (v is None) and (u is None or np.array_equal(u, v)), (a.name, f)
