"""Drop-in for the reference's utils_graph_learning.py (imported by utils.py:5, models_graph_classification*.py:11 and
graph_filters/*)."""
from gsn_amd.encoding import DiscreteEmbedding, multi_embedding, one_hot_encoder, zero_encoder  # noqa: F401
from gsn_amd.layers import central_encoder, global_add_pool_sparse, global_mean_pool_sparse  # noqa: F401


def multi_class_accuracy(y_hat, y, reduction='sum'):
    """utils_graph_learning.py:11-20."""
    pred = y_hat.max(1)[1]
    if reduction == 'sum':
        return pred.eq(y).sum().float()
    if reduction == 'mean':
        return pred.eq(y).mean().float()
    raise NotImplementedError('Reduction {} not currently implemented.'.format(reduction))
