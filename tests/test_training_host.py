"""CPU checks of the training path's host side: the oracle's autograd gradients pinned against the REFERENCE's own
gradients (tests/golden/*.npz "grad/*"), training-mode edge dropout, the size queries of the training entry points,
and the graph-wise data-parallel gradient sync (world_size 2, gloo) with the oracle as the compute function."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from amber_amd import _lib, engine, sharding, synthetic as syn
from amber_amd.mpn.common.edge_dropout import EdgeDropout, dropout_edge
from conftest import golden_cases, load_golden, load_golden_grads, rel_err
from oracle import mpn_oracle as orc

TRAINABLE = [n for n in golden_cases() if load_golden_grads(n) is not None]
# same arithmetic, same library as the reference (torch CPU autograd): tight
ORACLE_GRAD_TOL = 1e-5


def _oracle_grads(cfg, params, x, edge_attr, edge_index, mask, grad_out):
    # spectral norm's power-iteration vectors are buffers, not parameters
    p = {k: v.clone().requires_grad_(not k.endswith(("weight_u", "weight_v"))) for k, v in params.items()}
    out = orc.supervised_mpn_forward(p, cfg, x, edge_attr, edge_index, mask)
    (out * grad_out).sum().backward()
    # no dependency (the outer LayerNorm of the last block's edges): autograd leaves None, the gradient is zero
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if v.requires_grad}


@pytest.mark.parametrize("name", TRAINABLE)
def test_oracle_autograd_matches_reference_gradients(name):
    cfg, params, d = load_golden(name)
    grad_out, want = load_golden_grads(name)
    got = _oracle_grads(cfg, params, d["x"], d["edge_attr"], d["edge_index"], d["mask_output"], grad_out)
    assert sorted(got) == sorted(want)  # every parameter of the reference has a gradient, by its state_dict name
    for k in want:
        assert rel_err(got[k], want[k]) <= ORACLE_GRAD_TOL, k


def test_trainable_fixtures_cover_the_built_configurations():
    kinds = set()
    for name in TRAINABLE:
        cfg, _, _ = load_golden(name)
        st = cfg["stack"]
        kinds.add((st["mlp"]["activation_function"], st["residual_connections"], st["layer_norm"], st["mlp"]["num_layers"],
                   cfg["in_node_features"] == 0, cfg["in_edge_features"] == 0, st.get("num_step_repeats", 1) > 1))
    assert any(k[0] == "tanh" and k[1] is None and k[2] is None for k in kinds)
    assert any(k[3] == 3 and k[6] for k in kinds) and any(k[4] for k in kinds) and any(k[5] for k in kinds)


def test_edge_dropout_train_and_eval():
    g = syn.grid_batch(num_graphs=1, nx=6, ny=6)
    ei, ea = g.edge_index.clone(), g.edge_attr.clone()
    drop = EdgeDropout(dropout_prob=0.4)
    drop.eval()
    drop(g)
    assert g.edge_index is not None and torch.equal(g.edge_index, ei) and torch.equal(g.edge_attr, ea)
    drop.train()
    torch.manual_seed(5)
    mask = torch.rand(ei.shape[1]) >= 0.4  # torch_geometric.utils.dropout_edge's draw
    torch.manual_seed(5)
    drop(g)
    assert torch.equal(g.edge_index, ei[:, mask]) and torch.equal(g.edge_attr, ea[mask])
    kept, m = dropout_edge(ei, p=0.0, training=True)
    assert kept is ei and bool(m.all())
    with pytest.raises(ValueError):
        dropout_edge(ei, p=1.5)


def test_training_size_queries_and_validation():
    lib = _lib.load()
    kw = dict(latent=64, num_steps=3, num_step_repeats=2, mlp_layers=2, activation="leakyrelu", residual="inner",
              layer_norm="inner", in_node_features=4, in_edge_features=2, has_head=True, decoder_layers=2)
    d = engine.make_desc(**kw)
    n, e = 1000, 7000
    a256 = lambda b: (b + 255) // 256 * 256  # noqa: E731
    t1 = 3 * 2 + 1  # block inputs of every step application + the final state
    # ... and the mean aggregate of every step application
    assert lib.amber_mpn_tape_bytes(C.byref(d), n, e) == a256(t1 * n * 64 * 4) + a256(t1 * e * 64 * 4) + a256((t1 - 1) * n * 64 * 4) + 256
    assert lib.amber_mpn_backward_workspace_bytes(C.byref(d), n, e) >= lib.amber_mpn_workspace_bytes(C.byref(d), n, e)
    # configurations the backward kernels do not take are refused by name, before any pointer is touched
    for bad, word in ((dict(decoder_norm="batch"), b"BatchNorm"),):
        db = engine.make_desc(**{**kw, **bad})
        rc = lib.amber_mpn_forward_train(C.byref(db), 0, 0, n, e, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        assert rc == -2 and word in lib.amber_mpn_last_error()
    # MLP embeddings and the decoder's LayerNorm need row buffers of their own
    dm = engine.make_desc(**{**kw, "embed_layers": 4, "decoder_norm": "layer"})
    assert lib.amber_mpn_backward_workspace_bytes(C.byref(dm), n, e) > lib.amber_mpn_backward_workspace_bytes(C.byref(d), n, e)
    rc = lib.amber_mpn_backward(C.byref(d), 0, 0, n, e, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"NULL" in lib.amber_mpn_last_error()


def test_training_on_cpu_tensors_fails_loudly():
    """No CPU path, in training as in inference."""
    cfg, params, d = load_golden("debug_L64_s2")
    from amber_amd.architecture.supervised_mpn import SupervisedMPN
    from amber_amd.config import AttrDict

    model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=d["x"], edge_attr=d["edge_attr"])).train()
    g = syn.GraphBatch(x=d["x"], edge_attr=d["edge_attr"], edge_index=d["edge_index"], batch=d["batch"])
    with pytest.raises(RuntimeError, match="no CPU path|MI355X"):
        model(g)


# ------------------------------------------------------------------------------------------------------
# gradient sync of graph-wise data parallelism: one all-reduce of the flat gradient buffer
# ------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(grads: dict, order):
    return torch.cat([grads[k].reshape(-1) for k in order])


def _worker(rank, world, port, name, mode, ret):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, params, d = load_golden(name)
    grad_out, _ = load_golden_grads(name)
    g = syn.GraphBatch(x=d["x"], edge_attr=d["edge_attr"], edge_index=d["edge_index"], batch=d["batch"], ptr=d["ptr"],
                       mask_output=d["mask_output"])
    sub = sharding.shard_batch(g, rank, world)
    n0, n1 = sub.node_range
    kept_before = int(d["mask_output"][:n0].sum())
    kept_here = int(d["mask_output"][n0:n1].sum())
    # the checker stands in for the device backward: this test is about the sync of the flat gradient buffer
    grads = _oracle_grads(cfg, params, sub.x, sub.edge_attr, sub.edge_index, sub.mask_output,
                          grad_out[kept_before:kept_before + kept_here])
    flat = _flat(grads, sorted(grads))
    engine.grad_sync = mode
    engine._sync_flat_gradient(flat)
    if rank == 0:
        ret["flat"] = flat.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sum", "mean"])
def test_world_size_2_gradient_sync_equals_unsharded(mode):
    """sum(out * grad_out) is additive over graphs: the summed per-rank gradients equal the unsharded gradient
    ("mean" = DistributedDataParallel's convention: the same divided by the world size)."""
    name = "hier_L64_s3"
    _, want = load_golden_grads(name)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), name, mode, ret), nprocs=2, join=True)
    ref = _flat(want, sorted(want)) / (2.0 if mode == "mean" else 1.0)
    assert rel_err(ret["flat"], ref) <= 1e-5


def test_gradient_sync_is_a_no_op_without_a_process_group():
    t = torch.ones(4)
    engine._sync_flat_gradient(t)
    assert torch.equal(t, torch.ones(4))
