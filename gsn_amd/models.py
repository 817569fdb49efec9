"""GNNSubstructures (models_graph_classification.py:14-246) and GNN_OGB (models_graph_classification_ogb_original.py:17-268,
the virtual-node model of BASELINE config 4) assembled from this package's modules -- SURVEY.md 8(f) row 3.

Same constructor arguments, same parameter / buffer names (``input_node_encoder``, ``edge_encoder.{i}``, ``id_encoder.{i}``,
``degree_encoder``, ``conv.{i}.*``, ``lin_proj.{i}.*``, ``batch_norms.{i}.*``) so reference checkpoints load, same forward.
What differs is where the work happens: the between-layer ``BatchNorm1d`` + activation (:226-228) is folded into the epilogue
of the layer's last dense stage (no extra pass over [N, d]), the readout is the segmented-sum kernel, encoders and jk
projections run on the HIP stages.  The reference's own model class also runs unchanged on the drop-in modules
(INTEGRATION.md); this one is the version without PyTorch compute between the kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import flags, layers
from .encoding import DiscreteEmbedding
from .layers import (Codes, GSN_edge_sparse, GSN_edge_sparse_ogb, GSN_sparse, MPNN_edge_sparse, MPNN_edge_sparse_ogb, MPNN_sparse,
                     add_by_graph, choose_activation, global_add_pool_sparse, global_mean_pool_sparse, mlp, run_linear_module)


def _register_partition(data, edge_index):
    """A batch object may carry ``graph_partition = (node_ptr, edge_ptr, max_nodes, max_edges[, check])`` (int64 [G + 1] pointers of the
    collated graphs): the layers then build their aggregation index with one launch per batch (layers.set_graph_partition) and
    the readout uses the node pointers as its segment bounds (layers.set_batch_partition).  ``check`` (default True) reads the build's
    status word back -- a host synchronisation; False for a forward that is captured into a HIP graph (gsn_amd.graphs)."""
    part = getattr(data, "graph_partition", None)
    if part is not None and edge_index.is_cuda:
        from . import layers as _layers
        _layers.set_graph_partition(edge_index, part[0], part[1], part[2], part[3], check=bool(part[4]) if len(part) > 4 else True)
        batch = getattr(data, "batch", None)
        if batch is not None and batch.is_cuda:
            _layers.set_batch_partition(batch, part[0])           # the readout's rows are grouped by graph already


_PARAMETER_FREE_ENCODERS = ("one_hot_encoder", "atom_one_hot_encoder", "bond_one_hot_encoder")


def _codes_of(enc, t):
    """``Codes`` standing for ``enc(t)`` when ``enc`` is a parameter-free one-hot encoder over integer codes on the GPU (the encoder's own
    class counts, no clamp, no read-back of the out-of-range flag: exactly what its dense rows would hold), else None."""
    if getattr(enc, "encoder_name", None) not in _PARAMETER_FREE_ENCODERS or not isinstance(t, torch.Tensor) or not t.is_cuda or t.is_floating_point():
        return None
    t = t.unsqueeze(-1) if t.dim() == 1 else t
    dims = list(enc.encoder.d_in)
    if t.dim() != 2 or t.shape[1] != len(dims):
        return None
    return Codes(t, dims, clamp=False, check=False)


def _encode_once(memo, enc, x, training):
    """The reference encodes ``data.identifiers`` / ``data.edge_features`` once per LAYER (models_graph_classification.py:213-223); the
    one-hot encoders have no parameters, so every layer's encoder returns the same rows: they are made once per forward.  An encoder
    with parameters is re-used only for the SAME module in eval mode (in train mode its BatchNorm statistics move with every call, as in
    the reference).  ``memo``: a dict that lives for one forward."""
    name = getattr(enc, "encoder_name", None)
    if name in _PARAMETER_FREE_ENCODERS:
        key = (name, tuple(enc.encoder.d_in), id(x))
    elif not training:
        key = (id(enc), id(x))
    else:
        return enc(x)
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = enc(x)
    return hit


def _sum_tables(enc):
    """The embedding tables of an encoder that is a plain sum of one table row per code column (multi_embedding with aggr 'sum', the ogb
    Atom / Bond encoders), else None."""
    name = getattr(enc, "encoder_name", None)
    e = getattr(enc, "encoder", None)
    if name == "embedding" and getattr(e, "aggr", None) == "sum":
        return [m.weight for m in e.encoder]
    if name in ("atom_encoder", "bond_encoder"):
        return [m.weight for m in getattr(e, e._list_name)]
    return None


def _integer_codes(t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.is_floating_point() or t.dim() not in (1, 2):
        return None
    return t.unsqueeze(-1) if t.dim() == 1 else t


def _fused_edge_encoding(memo, id_enc, ef_enc, identifiers, edge_features):
    """``id_enc(identifiers) + ef_enc(edge_features)`` in ONE launch when both encoders are sums of table rows of one width over integer codes
    with one row per edge (flags.FUSE_EDGE_ENCODERS); None when they are not."""
    if not flags.FUSE_EDGE_ENCODERS:
        return None
    ti, te = _sum_tables(id_enc), _sum_tables(ef_enc)
    ci, ce = _integer_codes(identifiers), _integer_codes(edge_features)
    if ti is None or te is None or ci is None or ce is None:
        return None
    if ci.shape[0] != ce.shape[0] or ci.shape[1] != len(ti) or ce.shape[1] != len(te) or len(ti) + len(te) > 16:
        return None
    if ti[0].shape[1] != te[0].shape[1] or ti[0].device != ci.device:
        return None
    key = ("edge codes", id(identifiers), id(edge_features))
    codes = memo.get(key)
    if codes is None:            # (the concatenated code columns: once per forward, every layer's encoders read them)
        codes = memo[key] = torch.cat([ci.to(torch.int64), ce.to(torch.int64)], 1).contiguous()
    from .encoding import embed_columns
    return embed_columns(codes, ti + te, False)


class GNNSubstructures(nn.Module):
    def __init__(self, in_features, out_features, encoder_ids, d_in_id, in_edge_features=None, d_in_node_encoder=None,
                 d_in_edge_encoder=None, encoder_degrees=None, d_degree=None, **kwargs):
        super().__init__()
        seed = kwargs["seed"]
        self.model_name = kwargs["model_name"]
        self.readout = kwargs["readout"] if kwargs["readout"] is not None else "sum"
        self.dropout_features = kwargs["dropout_features"]
        self.bn = kwargs["bn"]
        self.final_projection = kwargs["final_projection"]
        self.inject_ids = kwargs["inject_ids"]
        self.inject_edge_features = kwargs["inject_edge_features"]
        self.random_features = kwargs["random_features"]
        id_scope = kwargs["id_scope"]
        d_msg, d_out, d_h = kwargs["d_msg"], kwargs["d_out"], kwargs["d_h"]
        aggr = kwargs["aggr"] if kwargs["aggr"] is not None else "add"
        flow = kwargs["flow"] if kwargs["flow"] is not None else "target_to_source"
        msg_kind = kwargs["msg_kind"] if kwargs["msg_kind"] is not None else "general"
        train_eps = kwargs["train_eps"] if kwargs["train_eps"] is not None else [False for _ in range(len(d_out))]
        activation_mlp, bn_mlp, jk_mlp = kwargs["activation_mlp"], kwargs["bn_mlp"], kwargs["jk_mlp"]
        degree_embedding = kwargs["degree_embedding"] if kwargs["degree_as_tag"][0] else "None"
        degree_as_tag, retain_features = kwargs["degree_as_tag"], kwargs["retain_features"]
        enc_kw = {"seed": seed, "activation_mlp": activation_mlp, "bn_mlp": bn_mlp, "aggr": kwargs["multi_embedding_aggr"]}

        self.input_node_encoder = DiscreteEmbedding(kwargs["input_node_encoder"], in_features, d_in_node_encoder,
                                                    kwargs["d_out_node_encoder"], **enc_kw)
        d_in = self.input_node_encoder.d_out
        if self.random_features:
            self.r_d_out = d_out[0]
            d_in = d_in + self.r_d_out
        edge_enc, d_ef = [], []
        for i in range(len(d_out) if kwargs["inject_edge_features"] else 1):
            e = DiscreteEmbedding(kwargs["edge_encoder"], in_edge_features, d_in_edge_encoder,
                                  kwargs["d_out_edge_encoder"][i], **enc_kw)
            edge_enc.append(e)
            d_ef.append(e.d_out)
        self.edge_encoder = nn.ModuleList(edge_enc)
        id_enc, d_id = [], []
        for i in range(len(d_out) if kwargs["inject_ids"] else 1):
            e = DiscreteEmbedding(kwargs["id_embedding"], len(d_in_id), d_in_id, kwargs["d_out_id_embedding"], **enc_kw)
            id_enc.append(e)
            d_id.append(e.d_out)
        self.id_encoder = nn.ModuleList(id_enc)
        self.degree_encoder = DiscreteEmbedding(degree_embedding, 1, d_degree, kwargs["d_out_degree_embedding"], **enc_kw)
        d_degree = self.degree_encoder.d_out

        conv, bns, proj = [], [], []
        for i in range(len(d_out)):
            kw = {"d_in": d_in, "d_degree": d_degree, "degree_as_tag": degree_as_tag[i], "retain_features": retain_features[i],
                  "d_msg": d_msg[i], "d_up": d_out[i], "d_h": d_h[i],
                  "d_ef": d_ef[i] if self.inject_edge_features else d_ef[0], "seed": seed, "activation_name": activation_mlp,
                  "bn": bn_mlp, "aggr": aggr, "msg_kind": msg_kind, "eps": 0, "train_eps": train_eps[i], "flow": flow,
                  "edge_embedding": kwargs["edge_encoder"], "id_embedding": kwargs["id_embedding"],
                  "extend_dims": kwargs["extend_dims"]}
            use_ids = ((i > 0 and kwargs["inject_ids"]) or i == 0) and self.model_name in {"GSN_sparse", "GSN_edge_sparse"}
            use_efs = ((i > 0 and kwargs["inject_edge_features"]) or i == 0) and self.model_name in {"GSN_edge_sparse", "MPNN_edge_sparse"}
            if use_ids:
                fn = GSN_edge_sparse if use_efs else GSN_sparse
                kw["d_id"] = d_id[i] if self.inject_ids else d_id[0]
                kw["id_scope"] = id_scope
            else:
                fn = MPNN_edge_sparse if use_efs else MPNN_sparse
            conv.append(fn(**kw))
            if self.final_projection[i]:
                jk = mlp(d_in, out_features, d_h[i], seed, activation_mlp, bn_mlp) if jk_mlp else nn.Linear(d_in, out_features)
            else:
                jk = None
            proj.append(jk)
            bns.append(nn.BatchNorm1d(d_out[i]) if self.bn[i] else None)
            d_in = d_out[i]
        if self.final_projection[-1]:
            jk = mlp(d_in, out_features, d_h[-1], seed, activation_mlp, bn_mlp) if jk_mlp else nn.Linear(d_in, out_features)
        else:
            jk = None
        proj.append(jk)
        self.conv = nn.ModuleList(conv)
        self.lin_proj = nn.ModuleList(proj)
        self.batch_norms = nn.ModuleList(bns)
        if self.readout == "sum":
            self.global_pool = global_add_pool_sparse
        elif self.readout == "mean":
            self.global_pool = global_mean_pool_sparse
        else:
            raise ValueError("Invalid graph pooling type.")
        self.activation_name = kwargs["activation"]
        self.activation = choose_activation(kwargs["activation"])

    def forward(self, data, print_flag=False, return_intermediate=False):
        kwargs = {"degrees": self.degree_encoder(data.degrees)}
        memo = {}
        edge_index = data.edge_index
        _register_partition(data, edge_index)
        # Eval forward over integer codes with one-hot encoders everywhere in front of layer 0 (BASELINE configs[1]'s model): layer 0 takes
        # the CODES (layers.Codes) -- its inputs go straight into exact fp16 row packs (gsn_one_hot_pack16_hip) and the layer runs on
        # csrc/layer_rp.hip; the dense one-hot rows of x and of the identifiers are never written (those of the edge features still are, for
        # the layers behind).  Any layer that cannot use the codes densifies them itself: same rows as the encoders'.
        codes0 = None
        if (not self.training and flags.PACK16_LAYER and flags.FUSED_LAYER and not self.random_features and len(self.conv) > 0
                and not self.final_projection[0]):
            cx = _codes_of(self.input_node_encoder, data.x)
            ci = _codes_of(self.id_encoder[0], getattr(data, "identifiers", None))
            ce = _codes_of(self.edge_encoder[0], data.edge_features) if hasattr(data, "edge_features") else None
            if cx is not None and ci is not None and (ce is not None or not hasattr(data, "edge_features")):
                codes0 = (cx, ci, ce)
        x = codes0[0] if codes0 is not None else self.input_node_encoder(data.x)
        if self.random_features:
            r = torch.rand(size=(x.shape[0], self.r_d_out), device=x.device).float()
            x = torch.cat((x, r), 1)
        x_interm = [x]
        for i in range(len(self.conv)):
            if i == 0 and codes0 is not None:
                kwargs["identifiers"], kwargs["edge_features"] = codes0[1], codes0[2]
            else:
                kwargs["identifiers"] = _encode_once(memo, self.id_encoder[i] if self.inject_ids else self.id_encoder[0], data.identifiers, self.training)
                if hasattr(data, "edge_features"):
                    kwargs["edge_features"] = _encode_once(memo, self.edge_encoder[i] if self.inject_edge_features else self.edge_encoder[0], data.edge_features, self.training)
                else:
                    kwargs["edge_features"] = None
            # BatchNorm1d + activation of models_graph_classification.py:226-228 ride in the layer's last epilogue
            x = self.conv[i](x, edge_index, post_bn=self.batch_norms[i] if self.bn[i] else None,
                             post_act=self.activation_name, **kwargs)
            x_interm.append(x)
        prediction = 0
        for i in range(len(self.conv) + 1):
            if self.final_projection[i]:
                x_global = self.global_pool(x_interm[i], data.batch)
                jk = self.lin_proj[i]
                y = jk(x_global) if isinstance(jk, mlp) else run_linear_module(jk, x_global)
                prediction = prediction + F.dropout(y, p=self.dropout_features[i], training=self.training)
        if return_intermediate:
            if isinstance(x_interm[0], Codes):
                x_interm[0] = x_interm[0].dense()           # (the encoder's rows, for whoever asks for them)
            return prediction, x_interm
        return prediction


class GNN_OGB(nn.Module):
    """models_graph_classification_ogb_original.py:17-268: ogb-style layers, optional virtual node (one embedding per graph,
    added to every vertex before each layer and updated from the pooled layer input, :252-259), residuals, sum of the
    selected intermediate representations -> readout -> Linear."""

    def __init__(self, in_features, out_features, encoder_ids, d_in_id, in_edge_features=None, d_in_node_encoder=None,
                 d_in_edge_encoder=None, encoder_degrees=None, d_degree=None, **kwargs):
        super().__init__()
        import copy
        seed = kwargs["seed"]
        self.model_name = kwargs["model_name"]
        self.readout = kwargs["readout"] if kwargs["readout"] is not None else "sum"
        self.dropout_features = kwargs["dropout_features"]
        self.bn = kwargs["bn"]
        self.final_projection = kwargs["final_projection"]
        self.residual = kwargs["residual"]
        self.inject_ids = kwargs["inject_ids"]
        self.vn = kwargs["vn"]
        id_scope = kwargs["id_scope"]
        d_msg, d_out, d_h = kwargs["d_msg"], kwargs["d_out"], kwargs["d_h"]
        aggr = kwargs["aggr"] if kwargs["aggr"] is not None else "add"
        flow = kwargs["flow"] if kwargs["flow"] is not None else "target_to_source"
        msg_kind = kwargs["msg_kind"] if kwargs["msg_kind"] is not None else "general"
        train_eps = kwargs["train_eps"] if kwargs["train_eps"] is not None else [False for _ in range(len(d_out))]
        activation_mlp, bn_mlp = kwargs["activation_mlp"], kwargs["bn_mlp"]
        degree_embedding = kwargs["degree_embedding"] if kwargs["degree_as_tag"][0] else "None"
        degree_as_tag, retain_features = kwargs["degree_as_tag"], kwargs["retain_features"]
        enc_kw = {"seed": seed, "activation_mlp": activation_mlp, "bn_mlp": bn_mlp, "aggr": kwargs["multi_embedding_aggr"],
                  "features_scope": kwargs["features_scope"]}
        self.input_node_encoder = DiscreteEmbedding(kwargs["input_node_encoder"], in_features, d_in_node_encoder,
                                                    kwargs["d_out_node_encoder"], **enc_kw)
        d_in = self.input_node_encoder.d_out
        if self.vn:
            vn_kw = copy.deepcopy(enc_kw)
            vn_kw["init"] = "zeros"
            self.vn_encoder = DiscreteEmbedding(kwargs["input_vn_encoder"], 1, [1], kwargs["d_out_vn_encoder"], **vn_kw)
            d_in_vn = self.vn_encoder.d_out
        edge_enc, d_ef = [], []
        for i in range(len(d_out)):
            e = DiscreteEmbedding(kwargs["edge_encoder"], in_edge_features, d_in_edge_encoder, kwargs["d_out_edge_encoder"][i], **enc_kw)
            edge_enc.append(e)
            d_ef.append(e.d_out)
        self.edge_encoder = nn.ModuleList(edge_enc)
        id_enc, d_id = [], []
        for i in range(len(d_out) if kwargs["inject_ids"] else 1):
            e = DiscreteEmbedding(kwargs["id_embedding"], len(d_in_id), d_in_id, kwargs["d_out_id_embedding"], **enc_kw)
            id_enc.append(e)
            d_id.append(e.d_out)
        self.id_encoder = nn.ModuleList(id_enc)
        self.degree_encoder = DiscreteEmbedding(degree_embedding, 1, d_degree, kwargs["d_out_degree_embedding"], **enc_kw)
        d_degree = self.degree_encoder.d_out
        conv, bns, mlp_vn = [], [], []
        for i in range(len(d_out)):
            if i > 0 and self.vn:
                mlp_vn.append(mlp(d_in_vn, kwargs["d_out_vn"][i - 1], d_h[i], seed, activation_mlp, bn_mlp))
                d_in_vn = kwargs["d_out_vn"][i - 1]
            kw = {"d_in": d_in, "d_degree": d_degree, "degree_as_tag": degree_as_tag[i], "retain_features": retain_features[i],
                  "d_msg": d_msg[i], "d_up": d_out[i], "d_h": d_h[i], "seed": seed, "activation_name": activation_mlp,
                  "bn": bn_mlp, "aggr": aggr, "msg_kind": msg_kind, "eps": 0, "train_eps": train_eps[i], "flow": flow,
                  "d_ef": d_ef[i], "edge_embedding": kwargs["edge_encoder"], "id_embedding": kwargs["id_embedding"],
                  "extend_dims": kwargs["extend_dims"]}
            use_ids = ((i > 0 and kwargs["inject_ids"]) or i == 0) and self.model_name == "GSN_edge_sparse_ogb"
            if use_ids:
                fn = GSN_edge_sparse_ogb
                kw["d_id"] = d_id[i] if self.inject_ids else d_id[0]
                kw["id_scope"] = id_scope
            else:
                fn = MPNN_edge_sparse_ogb
            conv.append(fn(**kw))
            bns.append(nn.BatchNorm1d(d_out[i]) if self.bn[i] else None)
            d_in = d_out[i]
        self.conv = nn.ModuleList(conv)
        self.batch_norms = nn.ModuleList(bns)
        if kwargs["vn"]:
            self.mlp_vn = nn.ModuleList(mlp_vn)
        if self.readout == "sum":
            self.global_pool = global_add_pool_sparse
        elif self.readout == "mean":
            self.global_pool = global_mean_pool_sparse
        else:
            raise ValueError("Invalid graph pooling type.")
        if self.vn:
            if kwargs["vn_pooling"] == "sum":
                self.global_vn_pool = global_add_pool_sparse
            elif kwargs["vn_pooling"] == "mean":
                self.global_vn_pool = global_mean_pool_sparse
            else:
                raise ValueError("Invalid graph virtual node pooling type.")
        self.lin_proj = nn.Linear(d_out[-1], out_features)
        self.activation_name = kwargs["activation"]
        self.activation = choose_activation(kwargs["activation"])

    def forward(self, data, return_intermediate=False):
        kwargs = {"degrees": self.degree_encoder(data.degrees)}
        memo = {}
        edge_index = data.edge_index
        _register_partition(data, edge_index)
        if self.vn:
            n_graphs = layers.num_graphs_of(data.batch)            # (data.batch[-1] + 1 of the reference: sorted graph ids; one read per batch)
            vn_embedding = self.vn_encoder(torch.zeros(n_graphs, dtype=edge_index.dtype, device=edge_index.device))
        x = self.input_node_encoder(data.x)
        x_interm = [x]
        n_layers = len(self.conv)
        for i in range(n_layers):
            id_enc = self.id_encoder[i] if self.inject_ids else self.id_encoder[0]
            fused = None
            if isinstance(self.conv[i], GSN_edge_sparse_ogb) and self.conv[i].id_scope == "local" and hasattr(data, "edge_features"):
                fused = _fused_edge_encoding(memo, id_enc, self.edge_encoder[i], data.identifiers, data.edge_features)
            if fused is not None:
                # relu(x_j + (id_e + e_e)): the two encoders' rows summed by one launch, one per-edge stream into the layer
                kwargs["identifiers"], kwargs["edge_features"] = fused, None
            else:
                kwargs["identifiers"] = _encode_once(memo, id_enc, data.identifiers, self.training)
                kwargs["edge_features"] = _encode_once(memo, self.edge_encoder[i], data.edge_features, self.training) if hasattr(data, "edge_features") else None
            if self.vn:
                x_interm[i] = add_by_graph(x_interm[i], vn_embedding, data.batch)
            last = i == n_layers - 1
            # BatchNorm1d (+ activation on all but the last layer) fused into the layer's last stage (:241-246)
            x = self.conv[i](x_interm[i], edge_index, post_bn=self.batch_norms[i] if self.bn[i] else None,
                             post_act="identity" if last else self.activation_name, **kwargs)
            x = F.dropout(x, self.dropout_features[i], training=self.training)
            if self.residual:
                x = x + x_interm[-1]
            x_interm.append(x)
            if not last and self.vn:
                vn_temp = self.global_vn_pool(x_interm[i], data.batch) + vn_embedding
                # the activation after mlp_vn rides in its last stage unless the residual form needs the raw output
                if self.residual:
                    vn_embedding = self.mlp_vn[i](vn_temp)
                    vn_embedding = vn_embedding + F.dropout(self.activation(vn_embedding), self.dropout_features[i], training=self.training)
                else:
                    vn_embedding = F.dropout(self.mlp_vn[i](vn_temp, post=(None, self.activation_name)),
                                             self.dropout_features[i], training=self.training)
        prediction = 0
        for i in range(n_layers + 1):
            if self.final_projection[i]:
                prediction = prediction + x_interm[i]
        x_global = self.global_pool(prediction, data.batch)
        out = run_linear_module(self.lin_proj, x_global)
        if return_intermediate:
            return out, x_interm
        return out
