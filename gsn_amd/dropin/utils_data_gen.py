"""Drop-in for the reference's utils_data_gen.py (imported by utils.py:4): ``generate_dataset`` with batched counting.
Prepared graphs are PyG ``Data`` objects when torch_geometric is installed (so the reference's DataLoader collates
them), otherwise gsn_amd.data.Data."""
import functools

from gsn_amd import dataset as _ds

try:
    from torch_geometric.data import Data as _Data
except ImportError:  # no PyG on this box: our own attribute bag
    from gsn_amd.data import Data as _Data


@functools.wraps(_ds.generate_dataset)
def generate_dataset(*args, **kwargs):
    kwargs.setdefault("data_cls", _Data)
    return _ds.generate_dataset(*args, **kwargs)


def _prepare(data, subgraph_dicts, subgraph_params, regression, dataset_name, ex_fn, cnt_fn):
    """Per-graph form (utils_data_gen.py:86-108); a one-graph call of the batched driver."""
    return _ds.prepare_graphs([data], subgraph_dicts, subgraph_params, regression, dataset_name, cnt_fn, data_cls=_Data)[0]
