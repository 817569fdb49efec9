"""Host logic of the exact fp16 row packs (gsn_amd/packs.py) on CPU tensors -- no kernel runs: who owns which columns of a pack, when a tag is
current (version counter, claimed columns, shapes), how a layer call's inputs are matched to ONE node pack and ONE edge pack."""
import torch

from gsn_amd import packs


def _tagged(rows, width, pack, col0):
    t = torch.zeros(rows, width)
    packs.claim(t, pack, col0)
    return t


def test_tag_lifecycle_version_and_release():
    npk = torch.zeros(10, packs.NODE_COLS, dtype=torch.float16)
    x = _tagged(10, 28, npk, 0)
    assert packs.tag_of(x, 10, packs.NODE_COLS) == (npk, 0)
    assert packs.tag_of(x, 11, packs.NODE_COLS) is None            # another row count
    x.add_(1.0)                                                    # an in-place write moves the version counter: the pack no longer describes x
    assert packs.tag_of(x, 10, packs.NODE_COLS) is None
    y = _tagged(10, 28, npk, 0)
    packs.release(y)                                               # a raw-pointer rewrite is announced by its producer
    assert packs.tag_of(y, 10, packs.NODE_COLS) is None


def test_columns_claimed_by_another_tensor_end_the_tag():
    epk = torch.zeros(7, packs.EDGE_COLS, dtype=torch.float16)
    ids = _tagged(7, 12, epk, 0)
    ef = _tagged(7, 4, epk, 12)
    assert packs.tag_of(ids, 7, packs.EDGE_COLS) == (epk, 0) and packs.tag_of(ef, 7, packs.EDGE_COLS) == (epk, 12)
    other = _tagged(7, 6, epk, 10)                                 # overlaps both
    assert packs.tag_of(ids, 7, packs.EDGE_COLS) is None and packs.tag_of(ef, 7, packs.EDGE_COLS) is None
    assert packs.tag_of(other, 7, packs.EDGE_COLS) == (epk, 10)
    again = _tagged(7, 12, epk, 0)                                 # the same columns re-claimed by a new tensor: the newest owner wins
    assert packs.tag_of(again, 7, packs.EDGE_COLS) == (epk, 0) and packs.tag_of(other, 7, packs.EDGE_COLS) is None


def test_lookup_matches_one_node_pack_and_one_edge_pack_in_concatenation_order():
    npk = torch.zeros(5, packs.NODE_COLS, dtype=torch.float16)
    epk = torch.zeros(9, packs.EDGE_COLS, dtype=torch.float16)
    x = _tagged(5, 28, npk, 0)
    ids, ef = _tagged(9, 12, epk, 0), _tagged(9, 4, epk, 12)
    assert packs.lookup(x, [ids, ef]) is None                      # a [N, 32] pack that no node-pack producer marked: column 31 may not be 1.0
    npk._gsn_node_pack = True                                      # (what node_pack / pack_node_codes set behind writing column 31)
    assert packs.lookup(x, [ids, ef]) == (npk, epk)
    assert packs.lookup(x, []) == (npk, None)                      # cat(x_i, x_j) alone
    assert packs.lookup(x, [ef, ids]) is None                      # wrong order: ef would have to start at column 0
    epk2 = torch.zeros(9, packs.EDGE_COLS, dtype=torch.float16)
    ef2 = _tagged(9, 4, epk2, 12)
    assert packs.lookup(x, [ids, ef2]) is None                     # two different edge packs
    assert packs.lookup(torch.zeros(5, 28), [ids, ef]) is None     # untagged x
    assert packs.lookup(x, [ids, torch.zeros(9, 4)]) is None       # one untagged per-edge tensor
    x_off = _tagged(5, 20, npk, 4)
    assert packs.lookup(x_off, [ids, ef]) is None                  # the node block must start at column 0


def test_tag_dies_with_its_tensor():
    import gc
    epk = torch.zeros(3, packs.EDGE_COLS, dtype=torch.float16)
    t = _tagged(3, 8, epk, 0)
    assert 0 in epk._gsn_owners
    del t
    gc.collect()
    assert epk._gsn_owners[0][0]() is None                         # a dead weak reference never matches a live tensor
    u = torch.zeros(3, 8)
    u._gsn_pack16 = (epk, 0, u._version, 0)                        # a forged tag without a claim
    assert packs.tag_of(u, 3, packs.EDGE_COLS) is None


class _FakeCodes:
    """the slice of gsn_amd.layers.Codes the ownership logic reads (a real Codes needs a GPU tensor)"""
    def __init__(self, rows, n_classes):
        self.codes = torch.zeros(rows, len(n_classes), dtype=torch.int64)
        self.n_classes = list(n_classes)
        self._pack16 = None


def test_codes_claims_enter_the_owner_table():
    """a Codes tag used to be checked against the code tensor's version only -- re-using a pack for another batch's codes left the
    first Codes object believing its encoding was still in the pack."""
    epk = torch.zeros(6, packs.EDGE_COLS, dtype=torch.float16)
    a, b = _FakeCodes(6, [4]), _FakeCodes(6, [4])
    packs._claim_codes(a, epk, 12)
    assert packs._codes_tag(a) is not None and packs._codes_tag(a)[:2] == (epk, 12)
    packs._claim_codes(b, epk, 12)                                 # the same columns encoded again from other codes
    assert packs._codes_tag(a) is None and packs._codes_tag(b)[:2] == (epk, 12)
    ids = _tagged(6, 14, epk, 0)                                   # a tensor claiming columns 0 .. 13 overlaps b's 12 .. 15
    assert packs._codes_tag(b) is None and packs.tag_of(ids, 6, packs.EDGE_COLS) == (epk, 0)
    c = _FakeCodes(6, [4])
    packs._claim_codes(c, epk, 12)                                 # ... and the other way round
    assert packs.tag_of(ids, 6, packs.EDGE_COLS) is None
    c.codes.add_(1)                                                # codes rewritten in place: the version counter moved
    assert packs._codes_tag(c) is None


def test_tags_made_before_a_capture_epoch_do_not_count_inside_it():
    """drop_input_caches() (called in front of a stream capture) only cleared the CSR cache; a persistent Codes object or fp32 tensor
    tagged during the warm-up kept a valid tag, the capture skipped the encoder launch, and a replay after ``codes.copy_(new)`` read the warm-up's
    rows.  Tags carry the epoch they were made in; the epoch moves with drop_input_caches()."""
    from gsn_amd import _caches
    epk = torch.zeros(4, packs.EDGE_COLS, dtype=torch.float16)
    t = _tagged(4, 8, epk, 0)
    cd = _FakeCodes(4, [4])
    packs._claim_codes(cd, epk, 12)
    assert packs.tag_of(t, 4, packs.EDGE_COLS) == (epk, 0) and packs._codes_tag(cd) is not None
    _caches.drop_input_caches()
    assert packs.tag_of(t, 4, packs.EDGE_COLS) is None and packs._codes_tag(cd) is None
    packs.claim(t, epk, 0)                                         # made again in the new epoch (inside the capture: by a recorded launch)
    packs._claim_codes(cd, epk, 12)
    assert packs.tag_of(t, 4, packs.EDGE_COLS) == (epk, 0) and packs._codes_tag(cd) is not None


def test_switches_set_on_the_layers_module_reach_gsn_amd_flags():
    """the switches moved from gsn_amd.layers to gsn_amd.flags; ``layers.FUSED_LAYER = False`` in existing user code must still switch."""
    from gsn_amd import flags, layers
    old = flags.FUSED_LAYER
    try:
        layers.FUSED_LAYER = not old
        assert flags.FUSED_LAYER == (not old) and layers.FUSED_LAYER == (not old) and "FUSED_LAYER" not in vars(layers)
        flags.KERNEL_TIMER = {}
        assert layers.KERNEL_TIMER == {}
    finally:
        flags.FUSED_LAYER, flags.KERNEL_TIMER = old, None
    import pytest
    with pytest.raises(AttributeError):
        layers.NO_SUCH_SWITCH
