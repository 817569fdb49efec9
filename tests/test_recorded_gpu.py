"""Second and later eval forwards of the `general` layers run a RECORDED launch (gsn_amd.layers._hip_recorded: the first forward's stage
descriptors / BatchNorm vectors / prepared weights with this call's pointers; models_graph_classification.py:204-247 calls four layers per
forward and at the reference's batch sizes the Python in front of a launch was 3-4 x the kernel).  Same rows as the full path, bit for bit; a
parameter / buffer / switch that moves, other input kinds or a batch outside the recorded kernel take the full path again."""
import networkx as nx
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CTOR0 = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
             d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
CTORW = dict(d_in=128, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
             d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")


def _batch(n_graphs, seed, dev):
    from gsn_amd import synth
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return b, t(b.node_ptr), t(b.edge_ptr), t(b.edge_index), t(b.atom_type), t(b.bond_type)


def _full_path(layer, *args, **kw):
    """the forward with every record dropped first"""
    from gsn_amd import layers
    layers.invalidate_caches(layer)
    with torch.no_grad():
        return layer(*args, **kw)


def test_layer0_on_codes_recorded_launch_equals_the_full_path_and_follows_the_parameters():
    from gsn_amd import layers
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**CTOR0).to(dev).eval()
    outs = []
    for seed, n_graphs in ((1, 40), (2, 40), (3, 77)):                   # other batches, another size: pointers and row counts follow
        b, node_ptr, edge_ptr, ei, atoms, bonds = _batch(n_graphs, seed, dev)
        ids = torch.randint(0, 3, (b.num_edges, 4), device=dev)
        mk = lambda: dict(identifiers=layers.Codes(ids, [3, 3, 3, 3], clamp=True), degrees=torch.zeros(b.num_nodes, device=dev),
                          edge_features=layers.Codes(bonds, [4]))
        with torch.no_grad():
            y = layer(layers.Codes(atoms, [28]), ei, **mk())
        assert getattr(layer, "_fplan", None) is not None and layer._fplan[1] == "pack16"
        with torch.no_grad():
            y2 = layer(layers.Codes(atoms, [28]), ei, **mk())             # (recorded)
        y_full = _full_path(layer, layers.Codes(atoms, [28]), ei, **mk())
        assert torch.equal(y, y_full) and torch.equal(y2, y_full)
        outs.append(y_full)
    # a parameter moves (optimizer-style in-place update): the record is refused, the new rows are those of the full path
    with torch.no_grad():
        layer(layers.Codes(atoms, [28]), ei, **mk())
        layer.msg_fn.fc[0].weight.mul_(1.25)
        y3 = layer(layers.Codes(atoms, [28]), ei, **mk())
    assert torch.equal(y3, _full_path(layer, layers.Codes(atoms, [28]), ei, **mk())) and not torch.equal(y3, outs[-1])
    # a BatchNorm buffer moves
    with torch.no_grad():
        layer(layers.Codes(atoms, [28]), ei, **mk())
        layer.update_fn.bn[0].running_mean.add_(0.3)
        y4 = layer(layers.Codes(atoms, [28]), ei, **mk())
    assert torch.equal(y4, _full_path(layer, layers.Codes(atoms, [28]), ei, **mk())) and not torch.equal(y4, y3)


@pytest.mark.parametrize("partition", [True, False])
def test_wide_layer_recorded_launch(partition):
    """d = 128 layers: with the batch's graph boundaries registered the graph-aligned kernel is recorded (csrc/layer_g.hip); without them the
    forward stays on the full path (row exponents are handed from layer to layer there) -- same rows either way."""
    from gsn_amd import layers
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    layer = layers.GSN_edge_sparse(**CTORW).to(dev).eval()
    post_bn = torch.nn.BatchNorm1d(128).to(dev).eval()
    with torch.no_grad():
        post_bn.running_mean.normal_(0, 0.1); post_bn.running_var.uniform_(0.5, 1.5)
    for seed in (5, 6):
        b, node_ptr, edge_ptr, ei, atoms, bonds = _batch(50, seed, dev)
        if partition:
            layers.set_graph_partition(ei, node_ptr, edge_ptr, int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max()), check=True)
        x = torch.randn(b.num_nodes, 128, device=dev)
        ids = torch.nn.functional.one_hot(torch.randint(0, 3, (b.num_edges, 4), device=dev), 3).reshape(-1, 12).float()
        ef = torch.nn.functional.one_hot(bonds, 4).float()
        kw = dict(identifiers=ids, degrees=torch.zeros(b.num_nodes, device=dev), edge_features=ef, post_bn=post_bn, post_act="relu")
        with torch.no_grad():
            y1 = layer(x, ei, **kw)
            y2 = layer(x, ei, **kw)
        if partition:
            assert layer._fplan[1] == "graphs"
        y_full = _full_path(layer, x, ei, **kw)
        assert torch.equal(y1, y_full) and torch.equal(y2, y_full)
    # the post-stage's BatchNorm moves: refused, recorded again
    with torch.no_grad():
        post_bn.running_var.mul_(2.0)
        y3 = layer(x, ei, **kw)
    assert torch.equal(y3, _full_path(layer, x, ei, **kw)) and not torch.equal(y3, y_full)


def test_switches_and_train_mode_leave_the_record():
    from gsn_amd import flags, layers
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    layer = layers.GSN_edge_sparse(**CTOR0).to(dev).eval()
    b, node_ptr, edge_ptr, ei, atoms, bonds = _batch(30, 9, dev)
    ids = torch.randint(0, 3, (b.num_edges, 4), device=dev)
    mk = lambda: dict(identifiers=layers.Codes(ids, [3, 3, 3, 3], clamp=True), degrees=torch.zeros(b.num_nodes, device=dev), edge_features=layers.Codes(bonds, [4]))
    with torch.no_grad():
        y = layer(layers.Codes(atoms, [28]), ei, **mk())
        flags.PACK16_LAYER = False
        try:
            y_np = layer(layers.Codes(atoms, [28]), ei, **mk())           # (the fp32-row kernel: another arithmetic path, same rows to rounding)
        finally:
            flags.PACK16_LAYER = True
        assert float((y - y_np).abs().max()) <= 1e-5 * float(y.abs().max())
        layer.train()
        y_tr = layer(layers.Codes(atoms, [28]), ei, **mk())               # train-mode BatchNorm: batch statistics, never the recorded launch
        layer.eval()
        y_again = layer(layers.Codes(atoms, [28]), ei, **mk())
    assert not torch.equal(y_tr, y)
    # (the train-mode forward moved the running statistics: the eval rows after it are new ones -- those of the full path on the new buffers)
    assert torch.equal(y_again, _full_path(layer, layers.Codes(atoms, [28]), ei, **mk())) and not torch.equal(y_again, y)
