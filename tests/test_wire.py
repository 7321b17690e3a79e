"""CPU tests of the canonical proof bytes and the `show` rendering (bulletproofs_amd/csrc/wire_host.hip)
against the oracle's independent restatement of the same layout; no GPU involved."""
import random

import pytest

import bp_oracle as o
from bulletproofs_amd import _lib
from util import golden, load_proof

R, P = o.R, o.P


def proofs():
    g = golden()
    pr = load_proof(g["bench_proof"]["proof"])
    small = load_proof(g["multi_proof"]["proof"])
    with_o = dict(small, t1Commit=None, productProof=dict(small["productProof"],
                                                          lCommits=[None] + small["productProof"]["lCommits"][1:]))
    no_rounds = dict(pr, productProof=dict(lCommits=[], rCommits=[], l=5, r=R - 1))
    return [pr, small, with_o, no_rounds]


def test_serialize_matches_layout_and_round_trips():
    for pr in proofs():
        data = _lib.serialize_proof(pr)
        assert data == o.wire_encode(pr)
        assert _lib.deserialize_proof(data) == pr


def test_deserialize_is_strict():
    pr = proofs()[0]
    data = _lib.serialize_proof(pr)
    bad_cases = [data[:-1], data + b"\x00", b"", data[:200]]
    # scalar >= r, coordinate >= p, point off the curve, unknown tag
    bad_cases.append(R.to_bytes(32, "big") + data[32:])
    x_off = 96 + 1
    bad_cases.append(data[:x_off] + P.to_bytes(32, "big") + data[x_off + 32:])
    bad_cases.append(data[:x_off + 63] + bytes([data[x_off + 63] ^ 1]) + data[x_off + 64:])
    bad_cases.append(data[:96] + b"\x02" + data[97:])
    for bad in bad_cases:
        with pytest.raises(_lib.BpError):
            _lib.deserialize_proof(bad)
    with pytest.raises(_lib.BpError):
        _lib.serialize_proof(dict(pr, t=R))


@pytest.mark.parametrize("style", [0, 1])
def test_show_matches_derived_show(style):
    old = o.SHOW_STYLE
    o.SHOW_STYLE = "derived" if style == 0 else "plain"
    try:
        for pr in proofs():
            assert _lib.show_proof(pr, style) == o.show_range_proof(pr)
    finally:
        o.SHOW_STYLE = old


def test_show_text_shape():
    txt = _lib.show_proof(proofs()[2], 0).decode()
    assert txt.startswith("RangeProof {tBlinding = P ") and txt.endswith("}}")
    assert ", t1Commit = O, " in txt and "lCommits = [O,A (P " in txt
