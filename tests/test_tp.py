"""Tensor-parallel sharding (exllama_amd/tp.py) on CPU: the shards of every matrix tile the full matrix exactly (oracle
dequantisation of shard vs full), and two processes over gloo reproduce a layer's MLP and attention output projection with
the partition's collectives (all-reduce of the partial sums; gather mode for act-order matrices).  The GPU parity of whole
rank models is tests/test_tp_gpu.py."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from exllama_amd import synth, tp
from oracle.model_oracle import dequant_rows_f32


def _w(t, key):
    g = t.get(key + ".g_idx")
    return dequant_rows_f32(t[key + ".qweight"].numpy(), t[key + ".qzeros"].numpy(), t[key + ".scales"].numpy(),
                            None if g is None else g.numpy())


def _ckpt(preset, gs, act, layers=1, seed=3):
    dims = synth.PRESETS[preset]
    return dims, synth.config_dict(dims, num_layers=layers), synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=seed, num_layers=layers)


@pytest.mark.parametrize("preset,gs,act,world", [("tiny_hd128", 128, False, 2), ("tiny_hd128", 128, False, 4), ("tiny_hd128", 64, False, 2),
                                                 ("tiny_hd128_gqa", 128, False, 2), ("tiny_hd128", 128, True, 2)])
def test_shards_tile_the_full_matrices(preset, gs, act, world):
    dims, cfg, t = _ckpt(preset, gs, act)
    hd = dims.head_dim
    p = "model.layers.0."
    shards = [tp.shard_tensors(t, cfg, r, world) for r in range(world)]
    plans = [pl for _, pl in shards]
    # the plans partition heads, kv heads and intermediate columns without gaps or overlap
    assert [pl.heads for pl in plans][0][0] == 0 and plans[-1].heads[1] == dims.num_attention_heads
    assert all(plans[r].heads[1] == plans[r + 1].heads[0] and plans[r].inter[1] == plans[r + 1].inter[0] for r in range(world - 1))
    assert plans[0].inter[0] == 0 and plans[-1].inter[1] == dims.intermediate_size
    # act-order: down_proj's row map is folded into the column order of gate / up before the cut (every rank computes the same map)
    fold = plans[0].fold_maps.get(0)
    assert (fold is not None) == bool(act) and all(torch.equal(pl.fold_maps[0], fold) for pl in plans if act)
    fm = None if fold is None else fold.numpy()
    for name in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "mlp.gate_proj", "mlp.up_proj"):
        full = _w(t, p + name)
        if fm is not None and name.startswith("mlp."):
            full = full[:, fm]
        assert np.array_equal(np.concatenate([_w(s, p + name) for s, _ in shards], axis=1), full), name
    full = _w(t, p + "self_attn.o_proj")                                  # act-order: gather mode = cut by output columns
    assert np.array_equal(np.concatenate([_w(s, p + "self_attn.o_proj") for s, _ in shards], axis=1 if act else 0), full)
    full = _w(t, p + "mlp.down_proj")                                     # always by rows -- the sequential rows once the map is folded
    assert all((p + "mlp.down_proj.g_idx") not in s for s, _ in shards if act)
    assert np.array_equal(np.concatenate([_w(s, p + "mlp.down_proj") for s, _ in shards], axis=0), full if fm is None else full[fm])
    assert torch.equal(t[p + "mlp.gate_proj.qweight"], _ckpt(preset, gs, act)[2][p + "mlp.gate_proj.qweight"])   # the caller's checkpoint is untouched
    # replicated tensors are the same objects; the local config describes the local shapes
    assert torch.equal(torch.cat([s["lm_head.weight"] for s, _ in shards], dim=0), t["lm_head.weight"])     # vocabulary rows
    for s, pl in shards:
        assert s["model.embed_tokens.weight"] is t["model.embed_tokens.weight"]
        c = tp.shard_config_dict(cfg, pl)
        assert c["hidden_size"] == dims.hidden_size and c["head_dim"] == hd
        assert s[p + "self_attn.q_proj.qweight"].shape[1] == c["num_attention_heads"] * hd
        assert s[p + "mlp.gate_proj.qweight"].shape[1] == c["intermediate_size"]


def test_placeholder_g_idx_and_storage_independence():
    """(a) Checkpoints without act-order may still carry an all-zero g_idx (the reference and Ex4bitLinear treat it as absent): the
    row-split o_proj / down_proj shards must pass it through as zeros on EVERY rank -- rebasing it would hand ranks > 0 a constant
    negative index that make_q4 rejects.  (b) No shard tensor that make_q4 rewrites in place (qweight) shares storage with the
    caller's checkpoint, also with world == 1 where every slice is the whole tensor."""
    dims, cfg, t = _ckpt("tiny_hd128", 128, False)
    p = "model.layers.0."
    for name, K in (("self_attn.o_proj", dims.hidden_size), ("mlp.down_proj", dims.intermediate_size)):
        t[p + name + ".g_idx"] = torch.zeros(K, dtype=torch.int32)
    for world in (1, 2):
        for r in range(world):
            s, pl = tp.shard_tensors(t, cfg, r, world)
            for name in ("self_attn.o_proj", "mlp.down_proj"):
                g = s[p + name + ".g_idx"]
                assert g.numel() == s[p + name + ".qweight"].shape[0] * 8 and not bool((g != 0).any()), (world, r, name)
            for key, v in s.items():
                if key.endswith(".qweight") or key == "lm_head.weight":
                    assert v.untyped_storage().data_ptr() != t[key].untyped_storage().data_ptr(), (world, r, key)
    # a real sequential index IS rebased to the shard's first group
    gs = 128
    t[p + "mlp.down_proj.g_idx"] = (torch.arange(dims.intermediate_size) // gs).to(torch.int32)
    s1, pl1 = tp.shard_tensors(t, cfg, 1, 2)
    assert int(s1[p + "mlp.down_proj.g_idx"].min()) == 0


def test_bad_partitions_are_refused():
    dims, cfg, t = _ckpt("tiny_hd128", 128, False)
    with pytest.raises(ValueError):
        tp.shard_tensors(t, cfg, 0, 3)                                    # 4 heads over 3 ranks
    with pytest.raises(ValueError):
        tp.TPPlan({**cfg, "intermediate_size": 256, "num_attention_heads": 4}, 0, 4, 128)   # 2 blocks of 128 columns, 4 ranks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _worker(rank, world, port, act, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims, cfg, t = _ckpt("tiny_hd128", 128, act)
    local, plan = tp.shard_tensors(t, cfg, rank, world)
    comm = tp.TensorParallel(plan, dist)
    p = "model.layers.0."
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, dims.hidden_size)).astype(np.float32)
    attn_full = rng.standard_normal((5, dims.hidden_size)).astype(np.float32)     # stands for the attention output, head-major
    # MLP: column-parallel gate / up, then row-parallel down + all-reduce (act-order too: down_proj's map is folded into gate / up)
    a = _silu(x @ _w(local, p + "mlp.gate_proj")) * (x @ _w(local, p + "mlp.up_proj"))
    y = comm.all_reduce(torch.from_numpy(a @ _w(local, p + "mlp.down_proj"))).numpy()
    # o_proj over this rank's heads
    hd = plan.head_dim
    mine = attn_full[:, plan.heads[0] * hd:plan.heads[1] * hd]
    if act:
        o = comm.all_gather_last(torch.from_numpy(comm.all_gather_last(torch.from_numpy(mine)).numpy() @ _w(local, p + "self_attn.o_proj")),
                                 plan.hidden_sizes).numpy()
    else:
        o = comm.all_reduce(torch.from_numpy(mine @ _w(local, p + "self_attn.o_proj"))).numpy()
    y_ref = (_silu(x @ _w(t, p + "mlp.gate_proj")) * (x @ _w(t, p + "mlp.up_proj"))) @ _w(t, p + "mlp.down_proj")
    o_ref = attn_full @ _w(t, p + "self_attn.o_proj")
    dist.barrier()
    out.put((rank, float(np.abs(y - y_ref).max() / np.abs(y_ref).max()), float(np.abs(o - o_ref).max() / np.abs(o_ref).max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("act", [False, True])
def test_two_ranks_over_gloo_reproduce_mlp_and_o_proj(act):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, act, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [out.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, ey, eo in got:
        assert ey < 1e-5 and eo < 1e-5, (rank, ey, eo)                    # fp32 sums in a different order
