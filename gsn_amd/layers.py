"""Sparse GSN / MPNN message-passing layers on HIP kernels: host-side mirror of the reference's ``graph_filters``.

Classes, constructor arguments, ``forward(x, edge_index, **kwargs)`` signature, error behaviour and state_dict key
names follow graph_filters/GSN_sparse.py, GSN_edge_sparse.py, MPNN_sparse.py, MPNN_edge_sparse.py,
GSN_edge_sparse_ogb.py, MPNN_edge_sparse_ogb.py and models_misc.mlp of the reference, so reference checkpoints load
(`load_state_dict`) and ``models_graph_classification.py`` can instantiate them unchanged.

Forward pass = libgsn_hip.so kernels only:
  * gsn_csr_build_hip        target-sorted CSR of the batch (cached per edge_index)
  * gsn_mlp_chain_fwd_hip    one or two fused Linear(+BatchNorm1d)(+activation) stages, input rows gathered / concatenated on
                             the fly, optionally with the scatter-add fused into the epilogue (gsn_segsum_prepare_hip)
  * gsn_linear_fwd_hip       the same stage for shapes outside the fused kernel (any K / n_out, elu / tanh)
  * gsn_propagate_fwd_hip    the stand-alone scatter-add (and the gin / ogb message assembly)
  * gsn_code_stage_fwd_hip   first Linear over integer-coded inputs as a weight-row gather (``Codes``)
Restructuring that only changes fp32 rounding order (tolerance 1e-5, tests/test_layers_gpu.py): for
``msg_kind='general'`` the last Linear of ``msg_fn`` is applied after the sum aggregation,
``sum_e (W r_e + b) = W (sum_e r_e) + deg * b`` (SURVEY.md 7 "design notes for the MP kernels").
Backward (training) = kernels with their own adjoints composed under autograd: gsn_propagate_bwd_hip (scatter-add),
gsn_bn_act_bwd_hip + gsn_wgrad_hip + the forward kernel on W^T (dense stages), gsn_gather_cat_hip (edge rows of the
general layers).  A PyTorch re-computation remains only for gradients with BatchNorm in eval mode.
There is no CPU path: calling a layer on CPU tensors raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _abi, flags, packs
# The host side is split by concern -- _runtime (timers, zero arenas), _index (Codes, CSR, partitions, readouts, propagate), _caches
# (derived-weight caches and their validation), _dense (forward dispatch of the dense stages and one-launch layers), _autograd (native
# adjoints) -- and this module keeps the reference's classes.  The names below are what models, encoders, tests and bench.py have always
# found here; the switches live in gsn_amd.flags.
from ._runtime import _MAX_BLOCKS, _f32c, _need_cuda, _timed, _zeros, choose_activation
from ._index import (Codes, _AddByGraphFn, _CSR_CACHE, _PARTITION, _PropagateFn, _csr_for, _dense, add_by_graph, build_csr, build_csr_graphs,
                     global_add_pool_sparse, global_mean_pool_sparse, num_graphs_of, one_hot_identifiers, propagate,
                     set_batch_partition, set_graph_partition)
from ._caches import (_CAPTURE_CACHED, _async_validate, _module_fingerprint, _note_cache, drop_capture_caches, drop_input_caches,
                      invalidate_caches)
from ._dense import _Stage, _bn_resolve, _chain_fits, _launch_stages, _layer_fused, _linear_hip, run_stages
from ._autograd import (_DenseStagesFn, _FoldWeightsFn, _GatherCatFn, _HipWithNativeBackward, _HipWithTorchBackward, _code_stage_segsum,
                        _dense_native_ok, _run, _segment_sum_cols, run_stages_autograd)

__all__ = ["mlp", "central_encoder", "GSN_sparse", "GSN_edge_sparse", "MPNN_sparse", "MPNN_edge_sparse",
           "GSN_edge_sparse_ogb", "MPNN_edge_sparse_ogb", "build_csr", "propagate", "run_stages", "one_hot_identifiers",
           "global_add_pool_sparse", "global_mean_pool_sparse", "Codes", "run_linear_module", "invalidate_caches", "drop_input_caches", "drop_capture_caches", "set_graph_partition", "build_csr_graphs"]


class mlp(nn.Module):
    """models_misc.mlp (models_misc.py:18-59): Linear -> [BatchNorm1d] -> activation ... -> Linear, same attribute
    names (``fc``, ``bn``) so state dicts are interchangeable; forward runs on the HIP dense stages."""

    def __init__(self, in_features, out_features, d_k, seed, activation="elu", batch_norm=False):
        super().__init__()
        self.in_features, self.out_features, self.d_k, self.seed = in_features, out_features, d_k, seed
        self.activation_name, self.batch_norm = activation, batch_norm
        fc, bn = [], []
        d_in = [in_features]
        d_k = d_k + [out_features]
        for i in range(len(d_k)):
            fc.append(nn.Linear(d_in[i], d_k[i], bias=True))
            d_in = d_in + [d_k[i]]
            if self.batch_norm and i != len(d_k) - 1:
                bn.append(nn.BatchNorm1d(d_k[i]))
        self.fc = nn.ModuleList(fc)
        self.bn = nn.ModuleList(bn)
        self.activation = choose_activation(activation)

    def stages(self, blocks, upto=None, first_weight=None, first_bias=None, post=None):
        """The mlp as a list of _Stage; ``blocks`` feed the first Linear (optionally with a replaced first weight).
        ``post = (BatchNorm1d or None, activation name)`` is applied to the output of the last Linear (the model's
        between-layer BatchNorm + activation, models_graph_classification.py:226-228, fused into the stage epilogue)."""
        n = len(self.fc) if upto is None else upto
        out = []
        for i in range(n):
            last = i == len(self.fc) - 1
            w = self.fc[i].weight if (i > 0 or first_weight is None) else first_weight
            b = self.fc[i].bias if (i > 0 or first_bias is None) else first_bias
            bn = self.bn[i] if (self.batch_norm and not last) else None
            act = "identity" if last else self.activation_name
            if last and post is not None:
                bn, act = post
            out.append(_Stage(w, b, bn, act, blocks if i == 0 else ()))
        return out

    def hip_forward(self, blocks, m_rows, upto=None, first_weight=None, first_bias=None, csr=None, post=None):
        stages = self.stages(blocks, upto, first_weight, first_bias, post)
        if post is not None and post[0] is not None and post[0].training != self.training:
            raise RuntimeError("mlp and the fused BatchNorm1d must be in the same train/eval mode")
        return run_stages(stages, m_rows, self.training, csr=csr)

    # -- differentiable PyTorch twin (used to back-propagate through the dense stages)
    def torch_forward(self, x, upto=None, post=None):
        n = len(self.fc) if upto is None else upto
        for i in range(n):
            x = self.fc[i](x)
            if i != len(self.fc) - 1:
                if self.batch_norm:
                    b = self.bn[i]
                    x = F.batch_norm(x, None if self.training else b.running_mean, None if self.training else b.running_var,
                                     b.weight, b.bias, self.training, 0.0, b.eps)
                x = self.activation(x)
            elif post is not None:
                b, act = post
                if b is not None:
                    x = F.batch_norm(x, None if b.training else b.running_mean, None if b.training else b.running_var,
                                     b.weight, b.bias, b.training, 0.0, b.eps)
                x = choose_activation(act)(x)
        return x

    def forward(self, x, post=None):
        _need_cuda(x, "mlp input")
        native = _dense_native_ok(self.stages([(x, None)], post=post))
        if torch.is_grad_enabled() and self.training and native:
            return run_stages_autograd(self.stages([(x, None)], post=post), x.shape[0], self.training)
        # eval mode: the fused forward; if a gradient is asked for after all, the stages are re-run on the kernels that keep what
        # their HIP adjoints need (no PyTorch twin: BatchNorm on running statistics is a per-column affine map there)
        extra = list(post[0].parameters()) if (post is not None and post[0] is not None) else []
        again = (lambda x_: run_stages_autograd(self.stages([(x_, None)], post=post), x_.shape[0], self.training)) if native and not self.training \
            else (lambda x_: self.torch_forward(x_, post=post))
        return _run(self, lambda: self.hip_forward([(x, None)], x.shape[0], post=post), again, [x], extra_params=extra,
                    native=native and not self.training)


def run_linear_module(lin, x):
    """A lone ``nn.Linear`` on the HIP dense stage (DiscreteEmbedding('linear'), utils_graph_learning.py:63-65)."""
    _need_cuda(x, "linear input")
    stage = lambda: run_stages([_Stage(lin.weight, lin.bias, None, "identity", [(x, None)])], x.shape[0], False)
    if flags.NATIVE_DENSE_BACKWARD:
        again = lambda x_: run_stages_autograd([_Stage(lin.weight, lin.bias, None, "identity", [(x_, None)])], x_.shape[0], False)
        return _run(lin, stage, again, [x], native=True)
    return _run(lin, stage, lambda x_: F.linear(x_, lin.weight, lin.bias), [x])


# ------------------------------------------------------------------------------------------------------------------
# central_encoder (utils_graph_learning.py:211-260): dummy "self loop" value of ids / edge features for GIN-style sums
# ------------------------------------------------------------------------------------------------------------------
class _SumEmbedding(nn.Module):
    """the reference's multi_embedding([1], d, aggr='sum') -> parameter name ``encoder.0.weight``"""

    def __init__(self, d_out):
        super().__init__()
        emb = nn.Embedding(1, d_out)
        torch.nn.init.xavier_uniform_(emb.weight.data)
        self.encoder = nn.ModuleList([emb])


class _DiscreteEmbeddingShim(nn.Module):
    def __init__(self, d_out):
        super().__init__()
        self.encoder = _SumEmbedding(d_out)


class central_encoder(nn.Module):
    def __init__(self, nb_encoder, d_ef, extend=True):
        super().__init__()
        self.extend, self.nb_encoder = extend, nb_encoder
        self.one_hot = "one_hot_encoder" in nb_encoder
        if self.one_hot:
            self.d_out = d_ef + 1 if extend else d_ef
        else:
            self.d_out = d_ef
            if extend:
                self.encoder = _DiscreteEmbeddingShim(d_ef)  # state key: encoder.encoder.encoder.0.weight

    def forward(self, x_nb, num_nodes):
        if self.one_hot and self.extend:
            x_nb = torch.cat((torch.zeros((x_nb.shape[0], 1), device=x_nb.device), x_nb), -1)
            x_central = torch.zeros((num_nodes, self.d_out), device=x_nb.device)
            x_central[:, 0] = 1.0
        elif (not self.one_hot) and self.extend:
            x_central = self.encoder.encoder.encoder[0].weight[0:1].expand(num_nodes, -1)
        else:
            x_central = torch.zeros((num_nodes, self.d_out), device=x_nb.device)
        return x_central, x_nb

    def central_row(self, device):
        """(the central value as ONE row [1][d_out], zero columns to put in front of the neighbours' block): what ``forward`` expands to
        num_nodes rows and concatenates -- gsn_propagate_self_fwd_hip takes it as a row stride of 0 and a column offset instead."""
        if (not self.one_hot) and self.extend:
            return self.encoder.encoder.encoder[0].weight[0:1], 0
        key = str(device)
        cache = self.__dict__.setdefault("_gsn_rows", {})
        if key not in cache:
            row = torch.zeros((1, self.d_out), device=device)
            if self.one_hot and self.extend:
                row[0, 0] = 1.0
            cache[key] = row
        return cache[key], (1 if (self.one_hot and self.extend) else 0)


# ------------------------------------------------------------------------------------------------------------------
# the layers
# ------------------------------------------------------------------------------------------------------------------
class _SparseLayer(nn.Module):
    """Shared implementation; subclasses fix (has_ids, has_ef, ogb)."""

    has_ids = True
    has_ef = False
    ogb = False

    def __init__(self, d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 d_ef=None, d_id=None, id_scope=None, aggr="add", msg_kind=None, eps=0, train_eps=False,
                 flow="source_to_target", **kwargs):
        super().__init__()
        if msg_kind is None:
            msg_kind = "ogb" if self.ogb else "general"
        if not self.ogb:
            d_msg = d_in if d_msg is None else d_msg
        self.flow, self.aggr, self.msg_kind = flow, aggr, msg_kind
        if self.has_ids:
            self.id_scope = id_scope
        self.degree_as_tag, self.retain_features = degree_as_tag, retain_features
        if degree_as_tag:
            d_in = d_in + d_degree if retain_features else d_degree
        d_id = d_id if self.has_ids else 0
        d_ef = d_ef if self.has_ef else 0
        name = self.__class__.__name__
        if self.ogb:
            if msg_kind != "ogb":
                raise NotImplementedError("msg kind {} is not currently supported.".format(msg_kind))
            self._init_eps(eps, train_eps)
            update_input_dim = d_in
        elif msg_kind == "gin":
            if self.has_ef:
                self.central_node_edge_encoder = central_encoder(kwargs["edge_embedding"], d_ef, extend=kwargs["extend_dims"])
                d_ef = self.central_node_edge_encoder.d_out
            if self.has_ids and id_scope == "local":
                self.central_node_id_encoder = central_encoder(kwargs["id_embedding"], d_id, extend=kwargs["extend_dims"])
                d_id = self.central_node_id_encoder.d_out
            self._init_eps(eps, train_eps)
            self.msg_fn = None
            update_input_dim = d_in + d_id + d_ef
        elif msg_kind == "general":
            local = (not self.has_ids) or id_scope == "local"
            msg_input_dim = 2 * d_in + d_id + d_ef if local else 2 * (d_in + d_id) + d_ef
            self.msg_fn = mlp(msg_input_dim, d_msg, d_h, seed, activation_name, bn)
            update_input_dim = d_in + d_msg
        else:
            raise NotImplementedError("msg kind {} is not currently supported.".format(msg_kind))
        self.update_fn = mlp(update_input_dim, d_up, d_h, seed, activation_name, bn)
        del name

    def _init_eps(self, eps, train_eps):
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        self.eps.data.fill_(self.initial_eps)

    # -- input preparation shared by both paths (reference: forward() prologue of every layer)
    def _prepare(self, x, kwargs):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if self.degree_as_tag:
            x = _dense(x)
        degrees = kwargs["degrees"]
        identifiers = kwargs["identifiers"] if self.has_ids or self.ogb else None
        if not self.has_ids:
            identifiers = None
        if degrees is not None:
            degrees = degrees.unsqueeze(-1) if degrees.dim() == 1 else degrees
        if self.degree_as_tag:
            x = torch.cat([x, degrees], -1) if self.retain_features else degrees
        ef = None
        if self.has_ef:
            ef = kwargs["edge_features"]
            if ef is None and not (self.ogb and identifiers is not None):
                raise RuntimeError("%s: edge_features missing" % type(self).__name__)
            # (None on an ogb layer with identifiers: the caller has added the edge-feature embedding into the identifier embedding,
            #  models._fused_edge_encoding -- relu(x_j + id_e + e_e) with the last two terms as one stream)
            if ef is not None:
                ef = ef.unsqueeze(-1) if ef.dim() == 1 else ef
        return x, identifiers, ef

    def forward(self, x, edge_index, **kwargs):
        if self.aggr == "mean":
            raise NameError("name 'aggr_index' is not defined")  # the reference's 'mean' branch is dead code (GSN_sparse.py:148)
        if self.aggr != "add":
            raise NotImplementedError("Aggregation kind {} is not currently supported.".format(self.aggr))
        if self.msg_kind not in ("gin", "general", "ogb"):
            raise NotImplementedError("Message kind {} is not currently supported.".format(self.msg_kind))
        post = None
        if kwargs.get("post_bn") is not None or kwargs.get("post_act") is not None:
            post = (kwargs.get("post_bn"), kwargs.get("post_act") or "identity")
        x, ids, ef = self._prepare(x, kwargs)
        _need_cuda(x, "x")
        if flags.VALIDATE_CACHES:           # (see _module_fingerprint: notices parameter writes that bypass tensor._version)
            fp = _module_fingerprint(self)
            if fp != getattr(self, "_gsn_fingerprint", None):
                invalidate_caches(self)
                self._gsn_fingerprint = fp
        elif flags.ASYNC_VALIDATE and not self.training and not torch.cuda.is_current_stream_capturing():
            _async_validate(self)
        # Row counts of the per-edge / per-vertex inputs: the reference fails in torch.cat / indexing when they do not fit
        # (GSN_sparse.py:118-132); the kernels would read past the tensors instead.
        n_rows, n_cols_e = x.shape[0], edge_index.shape[1]
        ids_per_edge = self.has_ids and (self.id_scope == "local")
        if ids is not None and ids.shape[0] != (n_cols_e if ids_per_edge else n_rows):
            raise RuntimeError("identifiers: %d rows, expected %d (id_scope %r: one row per %s)" % (
                ids.shape[0], n_cols_e if ids_per_edge else n_rows, self.id_scope, "edge" if ids_per_edge else "vertex"))
        if ef is not None and ef.shape[0] != n_cols_e:
            raise RuntimeError("edge_features: %d rows, expected one per edge (%d)" % (ef.shape[0], n_cols_e))
        _need_cuda(edge_index, "edge_index")
        given = [x, ids, ef]
        inputs = [t for t in given if t is not None and not isinstance(t, Codes)]

        def unpack(ts):     # float inputs come back from autograd, Codes are densified for the differentiable twin
            ts = list(ts)
            return tuple(None if t is None else (t.dense() if isinstance(t, Codes) else ts.pop(0)) for t in given)

        if torch.is_grad_enabled() and flags.NATIVE_DENSE_BACKWARD and self.training:
            # training: compositions of kernels that each have a HIP adjoint -- no PyTorch twin
            if self.ogb or self.msg_kind == "gin":   # propagate -> axpy -> update_fn
                return self._twin(edge_index, _dense(x), _dense(ids), _dense(ef), post=post, native=True)
            if self._general_native_ok():
                return self._general_train(edge_index, _dense(x), _dense(ids), _dense(ef), post)
        extra = list(post[0].parameters()) if (post is not None and post[0] is not None) else []
        if flags.NATIVE_DENSE_BACKWARD and not self.training:
            # eval mode: the fused forward keeps nothing; a gradient asked for after all re-runs the layer as the composition of kernels
            # that have HIP adjoints (the train-mode paths above; every BatchNorm1d in eval mode is an affine map there) -- no twin
            if self.ogb or self.msg_kind == "gin":
                again = lambda *ts: self._twin(edge_index, *unpack(ts), post=post, native=True)
            else:
                again = lambda *ts: self._general_train(edge_index, *unpack(ts), post)
            return _run(self, lambda: self._hip(edge_index, x, ids, ef, post), again, inputs, extra_params=extra, native=True)
        return _run(self, lambda: self._hip(edge_index, x, ids, ef, post), lambda *ts: self._twin(edge_index, *unpack(ts), post=post), inputs,
                    extra_params=extra)

    # -- message blocks ------------------------------------------------------------------------------------------
    def _sel(self):
        return 0 if self.flow == "target_to_source" else 1

    def _gin_parts(self, x, ids, ef, n):
        """(self parts, neighbour ids, neighbour ef, ids per node?) for the gin formulation"""
        self_parts = [x]
        ids_nb, ids_per_node = None, False
        if self.has_ids:
            if self.id_scope == "global":
                self_parts.append(ids); ids_nb, ids_per_node = ids, True
            else:
                c, ids_nb = self.central_node_id_encoder(ids, n)
                self_parts.append(c)
        ef_nb = None
        if self.has_ef:
            c, ef_nb = self.central_node_edge_encoder(ef, n)
            self_parts.append(c)
        return self_parts, ids_nb, ef_nb, ids_per_node

    def _self_plus_messages(self, edge_index, x, ids, ef):
        """(1 + eps) * self + sum of messages of the gin / ogb layers in ONE pass of the propagate kernel (forward: no elementwise
        tensor op, no concatenation; GSN_sparse.py:157-163, GSN_edge_sparse.py:95-109, GSN_edge_sparse_ogb.py:63-84 / :103-106)."""
        n, sel = x.shape[0], self._sel()
        if self.ogb:
            per_node = self.has_ids and self.id_scope == "global"
            return propagate(1, edge_index, sel, n, a=x, b=ids if self.has_ids else None, c=ef, b_per_node=per_node,
                             selfs=[x, ids] if per_node else [x], eps=self.eps)
        selfs, ids_nb, ef_nb, per_node, pads = [x], None, None, False, [0, 0]
        if self.has_ids:
            if self.id_scope == "global":
                selfs.append(ids); ids_nb, per_node = ids, True
            else:
                row, pads[0] = self.central_node_id_encoder.central_row(x.device)
                selfs.append(row); ids_nb = ids
        if self.has_ef:
            row, pad = self.central_node_edge_encoder.central_row(x.device)
            selfs.append(row); ef_nb = ef
            if ids_nb is None:      # (the edge features are then the kernel's block b)
                ids_nb, ef_nb, pads[0] = ef, None, pad
            else:
                pads[1] = pad
        return propagate(0, edge_index, sel, n, a=x, b=ids_nb, c=ef_nb, b_per_node=per_node, selfs=selfs, eps=self.eps, pads=pads)

    # -- recorded launch of the one-launch eval forward (r06) --------------------------------------------------------------
    def _fplan_key(self, raw, post):
        """What a recorded launch (gsn_amd._dense._layer_fused: stage descriptors, BatchNorm vectors, prepared weights) is valid for: the kinds and
        widths of the inputs, the post-stage, the switches, and the version counter + address of every parameter / buffer behind it."""
        # (looked up afresh on every call -- a replaced Parameter object is a different tensor --, by direct traversal of the two mlps: the
        #  generic parameters() / buffers() walk costs more than the launch)
        ts = []
        for m in (self.msg_fn, self.update_fn):
            for fc in m.fc:
                ts.append(fc.weight)
                if fc.bias is not None:
                    ts.append(fc.bias)
            if m.batch_norm:
                for bn in m.bn:
                    ts.extend(t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None)
                    if bn.training:
                        return None          # (a BatchNorm1d in train mode takes batch statistics: never the recorded eval launch)
        pb = post[0] if post is not None else None
        key = [tuple((isinstance(t, Codes), t.shape[1]) if t is not None else None for t in raw), None if post is None else (id(pb), post[1]),
               flags.FUSED_LAYER, flags.PACK16_LAYER, flags.GRAPH_ALIGNED_LAYER, flags.VALIDATE_CACHES, self.training]
        key.extend((t._version, t.data_ptr()) for t in ts)
        if pb is not None:
            key.append(pb.training)
            key.extend((t._version, t.data_ptr()) for t in (pb.weight, pb.bias, pb.running_mean, pb.running_var) if t is not None)
        return tuple(key)

    def _hip_recorded(self, edge_index, raw, n, E, sel, post):
        """Second and later eval forwards of a `general` layer: the launch recorded by the first one with this call's pointers -- no stage
        lists, no BatchNorm resolution, no descriptor building (at the reference's batch sizes that Python was 3-4 x the kernel's time:
        models_graph_classification.py:204-247 calls four layers per forward).  None: no valid record (first call, a parameter moved, other
        input kinds), or this batch is outside the recorded kernel -- the caller takes the full path, which records again."""
        plan = getattr(self, "_fplan", None)
        if plan is None or E == 0 or self.training or torch.cuda.is_current_stream_capturing():
            return None
        key = self._fplan_key(raw, post)
        if key is None or plan[0] != key:
            return None
        x, ids, ef = raw
        per_edge = [t for t in (ids if self.has_ids else None, ef if self.has_ef else None) if t is not None]
        if plan[1] == "pack16":
            if isinstance(x, Codes):
                pk = packs.from_codes(x, per_edge)
            else:
                pk = packs.lookup(x, per_edge)
            if pk is None:
                return None
            csr = _csr_for(edge_index, sel, n)
            x_ptr = pk[0].data_ptr() if isinstance(x, Codes) else x.data_ptr()
            if x_ptr % 16:
                return None
            return plan[2](x_ptr, n, edge_index.device, csr, [csr.tgt, csr.src] + [csr.perm] * len(per_edge), pk)
        if plan[1] == "graphs":
            if isinstance(x, Codes) or x.dtype != torch.float32 or not x.is_contiguous():
                return None
            csr = _csr_for(edge_index, sel, n)
            blocks = [(x, csr.tgt), (x, csr.src)]
            for t in per_edge:
                if isinstance(t, Codes):
                    t = t.dense()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    return None
                blocks.append((t, csr.perm))
            return plan[2](x, csr, blocks)
        return None

    # -- HIP forward ---------------------------------------------------------------------------------------------
    def _hip(self, edge_index, x, ids, ef, post=None):
        n = x.shape[0]
        sel = self._sel()
        E = edge_index.shape[1]
        raw = (x, ids, ef)
        record = None
        if (flags.FUSED_LAYER and not self.training and not self.ogb and self.msg_kind == "general" and len(self.msg_fn.fc) == 2
                and not (self.has_ids and self.id_scope != "local")):
            y = self._hip_recorded(edge_index, raw, n, E, sel, post)
            if y is not None:
                return y
            record = self._fplan_key(raw, post)
        self._fplan_record = record
        use_codes = (not self.ogb and self.msg_kind == "general" and len(self.msg_fn.fc) == 2 and E > 0
                     and all(isinstance(t, Codes) for t in raw if t is not None))
        # integer codes in, one launch: the one-hot encodings go straight into exact fp16 row packs (gsn_one_hot_pack16_hip: 64 / 32 bytes
        # per row, no fp32 one-hot tensor at all) and the layer runs on them (csrc/layer_rp.hip).  Identifiers may also be the tagged rows
        # of the counting kernel (gsn_amd.counting.count_batch(encoded_pack=...)).
        if (flags.PACK16_LAYER and flags.FUSED_LAYER and isinstance(raw[0], Codes) and not self.ogb and self.msg_kind == "general" and len(self.msg_fn.fc) == 2
                and E > 0 and not self.training and not (self.has_ids and self.id_scope != "local")
                and (raw[2] is None or isinstance(raw[2], Codes)) and (raw[1] is None or isinstance(raw[1], (Codes, torch.Tensor)))):
            y = self._fused_on_code_packs(edge_index, raw, n, E, sel, post)
            if y is not None:
                return y
        x = _f32c(_dense(x))
        if not use_codes:
            ids, ef = _dense(ids), _dense(ef)
        if self.ogb or self.msg_kind == "gin":
            xin = self._self_plus_messages(edge_index, x, ids, ef)
            return self.update_fn.hip_forward([(xin, None)], n, post=post)
        # general
        idx_i, idx_j = edge_index[sel].contiguous(), edge_index[1 - sel].contiguous()
        blocks = [(x, idx_i), (x, idx_j)]
        if self.has_ids:
            blocks += [(ids, None)] if self.id_scope == "local" else [(ids, idx_i), (ids, idx_j)]
        if self.has_ef:
            blocks.append((ef, None))
        mf = self.msg_fn
        uf = self.update_fn
        if len(mf.fc) >= 2:
            # all stages but the last Linear of msg_fn on E rows, then the sum aggregation S = sum_e r_e; the last Linear
            # commutes with the sum:  agg = W2 S + deg*b2, and it feeds update_fn's first Linear (weights [W3x | W3a]):
            #     [x | agg] W3^T = x W3x^T + S (W3a W2)^T + deg (W3a b2)^T
            # so it is folded into that Linear's weight (a [d_h x d_msg] by [d_msg x d_h] product, once per call).
            csr = _csr_for(edge_index, sel, n)
            s_agg = None
            if use_codes:
                # every input of msg_fn's first Linear is a one-hot code column: weight-row gather, no dense one-hot
                cblocks = [(raw[0], csr.tgt), (raw[0], csr.src)]
                if self.has_ids:
                    cblocks += [(raw[1], csr.perm)] if self.id_scope == "local" else [(raw[1], csr.tgt), (raw[1], csr.src)]
                if self.has_ef:
                    cblocks.append((raw[2], csr.perm))
                s_agg = _code_stage_segsum(mf, cblocks, csr, E)
                if s_agg is None:
                    ids, ef = _dense(ids), _dense(ef)
                    blocks = [(x, idx_i), (x, idx_j)]
                    if self.has_ids:
                        blocks += [(ids, None)] if self.id_scope == "local" else [(ids, idx_i), (ids, idx_j)]
                    if self.has_ef:
                        blocks.append((ef, None))
            # fused path: rows walked in target-sorted order; every block is gathered through ONE int32 index
            # (x_i: sorted target, x_j: sorted source, per-edge rows: perm), the scatter-add happens in the epilogue
            if s_agg is None and E > 0:
                sblocks = [(x, csr.tgt), (x, csr.src)]
                if self.has_ids:
                    sblocks += [(ids, csr.perm)] if self.id_scope == "local" else [(ids, csr.tgt), (ids, csr.src)]
                if self.has_ef:
                    sblocks.append((ef, csr.perm))
                # the whole layer in one launch where it fits (eval-mode BatchNorm, K_edge <= 80, widths <= 128)
                if post is None or post[0] is None or post[0].training == uf.training:
                    pk = None
                    if flags.PACK16_LAYER and not (self.has_ids and self.id_scope != "local"):
                        # exact fp16 packs of the inputs, when their producers left them (gsn_amd.packs): looked up on the caller's tensors
                        pk = packs.lookup(raw[0], [t for t in (raw[1] if self.has_ids else None, raw[2] if self.has_ef else None) if t is not None])
                    y = _layer_fused(x, csr, mf.stages(sblocks, upto=len(mf.fc) - 1),
                                     uf.stages([(x, None)], first_weight=self._folded_first_weight(x.shape[1]), post=post),
                                     self.training, owner=self, gen=getattr(self, "_fold_gen", 0), pack16=pk, record=self._fplan_record)
                    if y is not None:
                        return y
                # wide edge rows (K > 160: layers 1.. of a d = 128 model, K = 260): the node part of the Linear once per NODE,
                # the rest as a gather-add inside the scatter kernel
                s_agg = self._split_edge_stage(x, ids, ef, csr, n, E)
                if s_agg is None:
                    s_agg = mf.hip_forward(sblocks, E, upto=len(mf.fc) - 1, csr=csr)
            if s_agg is None:
                r = mf.hip_forward(blocks, E, upto=len(mf.fc) - 1)
                s_agg = propagate(0, edge_index, sel, n, b=r)
            w_first = self._folded_first_weight(x.shape[1])
            return uf.hip_forward([(x, None), (s_agg, None), (csr.deg4, None)], n, first_weight=w_first, post=post)
        msgs = mf.hip_forward(blocks, E)
        agg = propagate(0, edge_index, sel, n, b=msgs)
        return uf.hip_forward([(x, None), (agg, None)], n, post=post)


    def _fused_on_code_packs(self, edge_index, raw, n, E, sel, post):
        """The one-launch layer fed from integer codes (see _hip); None when the shapes are outside the packed-row kernel."""
        mf, uf = self.msg_fn, self.update_fn
        if post is not None and post[0] is not None and post[0].training != uf.training:
            return None
        per_edge = [t for t in (raw[1] if self.has_ids else None, raw[2] if self.has_ef else None) if t is not None]
        d_x = sum(raw[0].n_classes)
        if d_x > packs.NODE_COLS - 4 or sum(packs._width(t) for t in per_edge) > packs.EDGE_COLS:
            return None
        pk = packs.from_codes(raw[0], per_edge)
        if pk is None:
            return None
        csr = _csr_for(edge_index, sel, n)
        dev = edge_index.device
        # (shapes and block identities for the stage descriptors: the packed-row kernel does not read the fp32 pointers -- no kernel runs here)
        xph = torch.empty((n, d_x), dtype=torch.float32, device=dev)
        sblocks = [(xph, csr.tgt), (xph, csr.src)]
        for t in per_edge:
            sblocks.append((t if isinstance(t, torch.Tensor) else torch.empty((E, packs._width(t)), dtype=torch.float32, device=dev), csr.perm))
        return _layer_fused(xph, csr, mf.stages(sblocks, upto=len(mf.fc) - 1),
                            uf.stages([(xph, None)], first_weight=self._folded_first_weight(d_x), post=post),
                            self.training, owner=self, gen=getattr(self, "_fold_gen", 0), pack16=pk, pack_only=True, record=self._fplan_record)

    # -- differentiable `general` path on native adjoints ------------------------------------------------------------------
    def _general_native_ok(self):
        return self.training

    def _general_train(self, edge_index, x, ids, ef, post):
        """msg_fn's hidden stages on materialised edge rows -> sum per target -> [x | S | deg] through update_fn with the
        folded first weight (the fold itself is three tiny PyTorch matrix products, so it stays differentiable)."""
        n, sel = x.shape[0], self._sel()
        E = edge_index.shape[1]
        tensors, modes = [x, x], [sel, 1 - sel]
        if self.has_ids:
            if self.id_scope == "local":
                tensors.append(ids); modes.append(None)
            else:
                tensors += [ids, ids]; modes += [sel, 1 - sel]
        if self.has_ef:
            tensors.append(ef); modes.append(None)
        mf, uf = self.msg_fn, self.update_fn
        # (an edge-less batch walks the same graph with zero rows: every parameter and input then gets the ZERO gradient PyTorch gives it)
        # the edge rows cat(x_i, x_j, ids.., e) are read where they lie (gathered blocks), by the product and by its weight gradient
        eblocks, gather = [(t, None) for t in tensors], (edge_index, n, tuple(modes))
        if flags.GATHER_CAT_TRAIN or len(tensors) > _MAX_BLOCKS:      # (the assembled-rows form: A/B switch)
            eblocks, gather = [(_GatherCatFn.apply(edge_index, n, tuple(modes), *tensors), None)], None
        if len(mf.fc) < 2:      # a single Linear as msg_fn: nothing to fold (GSN_sparse.py:166-171 with d_h = [])
            msgs = run_stages_autograd(mf.stages(eblocks), E, True, gather=gather)
            agg = propagate(0, edge_index, sel, n, b=msgs)
            return run_stages_autograd(uf.stages([(x, None), (agg, None)], post=post), n, True)
        r = run_stages_autograd(mf.stages(eblocks, upto=len(mf.fc) - 1), E, True, gather=gather)
        s_agg = propagate(0, edge_index, sel, n, b=r)
        csr = _csr_for(edge_index, sel, n)
        last, w3 = mf.fc[-1], uf.fc[0].weight
        d_x = x.shape[1]
        if flags.FOLD_KERNEL and _FoldWeightsFn.takes(w3, last, d_x):
            # the fold  W3x | W3a W2 | W3a b2  and its adjoint: one launch each (gsn_fold_weights_{fwd,bwd}_hip)
            # (three zero columns behind the degree column: the degree block goes in four floats wide, csr.deg4, and the stage's rows are staged
            #  as float4 -- K = d_x + d_h + 1 is odd otherwise)
            w_first = _FoldWeightsFn.apply(w3, last.weight, last.bias, d_x, 3)
            stages = uf.stages([(x, None), (s_agg, None), (csr.deg4, None)], first_weight=w_first, post=post)
            return run_stages_autograd(stages, n, True)
        else:
            # ... as two dense stages with their own adjoints (rows = W3a; no library GEMM in the step)
            w3x, w3a = w3[:, :d_x], w3[:, d_x:].contiguous()
            w_fold = run_stages_autograd([_Stage(last.weight.t(), None, None, "identity", [(w3a, None)])], w3a.shape[0], True)
            b_fold = run_stages_autograd([_Stage(last.bias.unsqueeze(0), None, None, "identity", [(w3a, None)])], w3a.shape[0], True)
            w_first = torch.cat([w3x, w_fold, b_fold], 1)
        stages = uf.stages([(x, None), (s_agg, None), (csr.deg, None)], first_weight=w_first, post=post)
        return run_stages_autograd(stages, n, True)

    def _split_edge_stage(self, x, ids, ef, csr, n, E):
        """S[t] = sum_{e -> t} act(bn(cat(x_i, x_j, z_e) W1^T + b1)) with the node columns of W1 applied once per node:
        P = x [W_i | W_j]^T (gsn_linear_fwd_hip, N rows), then gsn_edge_split_sum_hip gathers P_i[t] + P_j[src] and adds
        z_e W_z^T per edge.  Eval-mode / no BatchNorm, one hidden edge stage, identity / relu, per-edge blocks of
        <= 16 columns; used when the concatenated row is wider than the fused chain kernels take (K > flags.SPLIT_EDGE_MIN_K).
        Returns None when the shape is outside that."""
        mf = self.msg_fn
        if not flags.SPLIT_EDGE_STAGE or len(mf.fc) != 2 or self.training or E == 0:
            return None
        st = mf.stages([], upto=1)[0]
        act = {"identity": 0, "relu": 1}.get(st.act)
        if act is None or (st.bn is not None and (st.bn.training or st.bn.running_mean is None)):
            return None
        d_x, d_h = x.shape[1], st.weight.shape[0]
        node_blocks, edge_blocks = [x], []
        if self.has_ids:
            (edge_blocks if self.id_scope == "local" else node_blocks).append(ids)
        if self.has_ef:
            edge_blocks.append(ef)
        d_n = sum(b.shape[1] for b in node_blocks)
        d_r = sum(b.shape[1] for b in edge_blocks)
        if 2 * d_n + d_r <= flags.SPLIT_EDGE_MIN_K or d_h % 4 or d_h > 256 or any(b.shape[1] % 4 for b in edge_blocks) \
                or d_r > (16 if d_h <= 128 else 8) or len(edge_blocks) > 2:
            return None
        _bn_resolve(st, None, E, False)
        w1, b1 = mf.fc[0].weight, mf.fc[0].bias
        bn_key = None if st.bn is None else tuple(t._version for t in (st.bn.running_mean, st.bn.running_var)) + \
            ((st.bn.weight._version, st.bn.bias._version) if st.bn.affine else ())
        key = (w1._version, w1.data_ptr(), None if b1 is None else b1._version, bn_key, d_x, d_n, d_r)
        cache = getattr(self, "_split_cache", None)
        if cache is None or cache[0] != key:
            w = w1.detach()
            scale = None
            bias = b1.detach() if b1 is not None else torch.zeros(d_h, device=w.device)
            if st.bn_params is not None:
                mean, scale, shift = st.bn_params
                bias = (bias - mean) * scale + shift
            # column layout of W1: x_i, x_j, then ids_i, ids_j (global scope) or ids (local), then edge features
            cols_i, cols_j, off = [w[:, :d_x]], [w[:, d_x:2 * d_x]], 2 * d_x
            if self.has_ids and self.id_scope != "local":
                d_id = ids.shape[1]
                cols_i.append(w[:, off:off + d_id]); cols_j.append(w[:, off + d_id:off + 2 * d_id]); off += 2 * d_id
            w_i, w_j, w_z = torch.cat(cols_i, 1), torch.cat(cols_j, 1), w[:, off:]
            if scale is not None:
                w_i, w_j, w_z = w_i * scale[:, None], w_j * scale[:, None], w_z * scale[:, None]
            w_n = torch.cat([w_i, w_j], 0).contiguous()                                   # [2 d_h, d_n]
            bias_n = torch.cat([bias, torch.zeros_like(bias)]).contiguous()               # the target half carries bias + BN shift
            wz_t = w_z.t().contiguous() if d_r else None                                  # [d_r, d_h]
            cache = (key, w_n, bias_n, wz_t)
            self._split_cache = cache
            _note_cache(self, "_split_cache")
        _, w_n, bias_n, wz_t = cache
        P = _linear_hip([(b, None) for b in node_blocks], w_n, bias_n, None, None, None, 0, n)          # [N, 2 d_h]
        zs = [_f32c(b) for b in edge_blocks]
        out = torch.empty((n, d_h), dtype=torch.float32, device=x.device)
        with _abi.device_guard(x.device), _timed("edge_split_sum", 4.0 * (E * (d_h + d_r) + 2.0 * n * d_h)):
            rc = _abi.lib().gsn_edge_split_sum_hip(n, E, csr.seg_ptr.data_ptr(), csr.src.data_ptr(), csr.perm.data_ptr(),
                                                   P.data_ptr(), P.data_ptr() + 4 * d_h, 2 * d_h,
                                                   zs[0].data_ptr() if zs else None, zs[0].shape[1] if zs else 0,
                                                   zs[1].data_ptr() if len(zs) > 1 else None, zs[1].shape[1] if len(zs) > 1 else 0,
                                                   _abi.ptr(wz_t), d_h, act, out.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_edge_split_sum_hip")
        return out

    def _folded_first_weight(self, d_x):
        """[W3x | W3a W2 | W3a b2] (see _hip); recomputed only when one of the three parameters changed."""
        mf, uf = self.msg_fn, self.update_fn
        last, w3p = mf.fc[-1], uf.fc[0].weight
        key = (last.weight._version, last.bias._version, w3p._version, last.weight.data_ptr(), w3p.data_ptr(), d_x)
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        w3 = w3p.detach()
        w3x, w3a = w3[:, :d_x], w3[:, d_x:].contiguous()
        w2t = last.weight.detach().t().contiguous()                       # [d_h_msg, d_msg]: rows = input index
        w_fold = _linear_hip([(w3a, None)], w2t, None, None, None, None, 0, w3a.shape[0])   # = W3a @ W2
        b_fold = _linear_hip([(w3a, None)], last.bias.detach().unsqueeze(0).contiguous(), None, None, None, None, 0, w3a.shape[0])
        # three zero columns after the degree column: the degree block is passed 4 floats wide (csr.deg4)
        w_first = torch.cat([w3x, w_fold, b_fold, torch.zeros_like(b_fold).expand(-1, 3)], 1).contiguous()
        self._fold_cache = (key, w_first)
        _note_cache(self, "_fold_cache")
        self._fold_gen = getattr(self, "_fold_gen", 0) + 1        # (a new tensor may reuse the old one's address)
        return w_first

    # -- differentiable twin (PyTorch ops + the HIP propagate with its own adjoint) ---------------------------------
    def _twin(self, edge_index, x, ids, ef, post=None, native=False):
        n = x.shape[0]
        sel = self._sel()
        if native and (self.ogb or self.msg_kind == "gin"):
            return self.update_fn(self._self_plus_messages(edge_index, x, ids, ef), post=post)
        if self.ogb:
            per_node = self.has_ids and self.id_scope == "global"
            agg = propagate(1, edge_index, sel, n, a=x, b=ids if self.has_ids else None, c=ef, b_per_node=per_node)
            self_msg = x + ids if per_node else x
            xin = (1 + self.eps) * self_msg + agg
            return self.update_fn(xin, post=post) if native else self.update_fn.torch_forward(xin, post=post)
        if self.msg_kind == "gin":
            self_parts, ids_nb, ef_nb, per_node = self._gin_parts(x, ids, ef, n)
            agg = propagate(0, edge_index, sel, n, a=x, b=ids_nb, c=ef_nb, b_per_node=per_node)
            xin = (1 + self.eps) * torch.cat(self_parts, -1) + agg
            return self.update_fn(xin, post=post) if native else self.update_fn.torch_forward(xin, post=post)
        idx_i, idx_j = edge_index[sel], edge_index[1 - sel]
        parts = [x[idx_i], x[idx_j]]
        if self.has_ids:
            parts += [ids] if self.id_scope == "local" else [ids[idx_i], ids[idx_j]]
        if self.has_ef:
            parts.append(ef)
        msgs = self.msg_fn.torch_forward(torch.cat(parts, -1))
        agg = propagate(0, edge_index, sel, n, b=msgs)
        return self.update_fn.torch_forward(torch.cat((x, agg), -1), post=post)

    def __repr__(self):
        if self.ogb:
            return "{}(update_fn = {})".format(self.__class__.__name__, self.update_fn)
        return "{}(msg_fn = {}, update_fn = {})".format(self.__class__.__name__, self.msg_fn, self.update_fn)


class GSN_sparse(_SparseLayer):
    """graph_filters/GSN_sparse.py:8-181 (GSN-v: id_scope='global', GSN-e: 'local'; no edge features)."""
    has_ids, has_ef, ogb = True, False, False

    def __init__(self, d_in, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        kwargs.pop("d_ef", None)
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class GSN_edge_sparse(_SparseLayer):
    """graph_filters/GSN_edge_sparse.py:8-175."""
    has_ids, has_ef, ogb = True, True, False

    def __init__(self, d_in, d_ef, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps,
                         flow=flow, **kwargs)


class MPNN_sparse(_SparseLayer):
    """graph_filters/MPNN_sparse.py (identifier-free twin of GSN_sparse)."""
    has_ids, has_ef, ogb = False, False, False

    def __init__(self, d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        kwargs.pop("d_ef", None)
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class MPNN_edge_sparse(_SparseLayer):
    """graph_filters/MPNN_edge_sparse.py (identifier-free twin of GSN_edge_sparse)."""
    has_ids, has_ef, ogb = False, True, False

    def __init__(self, d_in, d_ef, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class GSN_edge_sparse_ogb(_SparseLayer):
    """graph_filters/GSN_edge_sparse_ogb.py:9-134: msg = relu(x_j + id + e), GIN-style update."""
    has_ids, has_ef, ogb = True, True, True

    def __init__(self, d_in, d_ef, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="ogb", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps,
                         flow=flow, **kwargs)


class MPNN_edge_sparse_ogb(_SparseLayer):
    """graph_filters/MPNN_edge_sparse_ogb.py: msg = relu(x_j + e)."""
    has_ids, has_ef, ogb = False, True, True

    def __init__(self, d_in, d_ef, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="ogb", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


# The switches moved to gsn_amd.flags in r05 (FUSED_LAYER, PACK16_LAYER, KERNEL_TIMER, CODE_STATUS_CHECK, NATIVE_DENSE_BACKWARD, ...).  Code
# written against the earlier layout sets them HERE (``layers.FUSED_LAYER = False``): forwarded, both ways, so that such a switch still
# switches instead of creating a dead attribute.
import sys as _sys
import types as _types


class _LayersModule(_types.ModuleType):
    def __setattr__(self, name, value):
        if name.isupper() and not name.startswith("_") and hasattr(flags, name):
            setattr(flags, name, value)
            return
        super().__setattr__(name, value)

    def __getattr__(self, name):          # (only reached when the module itself has no such attribute)
        if name.isupper() and not name.startswith("_") and hasattr(flags, name):
            return getattr(flags, name)
        raise AttributeError("module %r has no attribute %r" % (self.__name__, name))


_sys.modules[__name__].__class__ = _LayersModule
