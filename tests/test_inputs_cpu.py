"""prepare_vlm_inputs / get_text_position_ids / collate padding — bit-exact against fixture F11, the outputs
of the reference functions driven by the same deterministic tokenizer stand-in (oracle/fake_tokenizer.py)."""
from pathlib import Path

import torch

from mmmm_amd.data.inputs import ConvTurn, collate_vlm_inputs, get_text_position_ids, prepare_vlm_inputs
from oracle.fake_tokenizer import FakeTokenizer

GOLD = torch.load(Path(__file__).parent / 'golden' / 'f11_vlm_inputs.pt', weights_only=False)


def _run(case, tok):
    return prepare_vlm_inputs([ConvTurn(q, a) for q, a in case['conv']], tok, case['n_img'], inference=case['inference'],
                              grounding=case['grounding'], max_seq_len=case['max_seq_len'], bop_weight=case['bop_weight'])


def test_prepare_vlm_inputs_is_bit_exact():
    tok = FakeTokenizer()
    for case, ref in zip(GOLD['cases'], GOLD['outputs']):
        inputs, text = _run(case, tok)
        assert text == ref['text']
        assert inputs.keys() == ref['inputs'].keys()
        for k, v in ref['inputs'].items():
            assert inputs[k].dtype == v.dtype and torch.equal(inputs[k], v), (case['conv'][0][0], k)


def test_collate_matches_reference_padding():
    tok = FakeTokenizer()
    train = [_run(c, tok)[0] for c in GOLD['cases'] if not c['inference']]
    got = collate_vlm_inputs(train)
    for k, v in GOLD['collated'].items():
        assert torch.equal(got[k], v), k


def test_text_positions_hold_at_phrase_tags():
    tok = FakeTokenizer()
    ids = torch.tensor(tok.encode('a <p>b c</p> d <p>e</p></p>'))
    pos = get_text_position_ids(ids, tok, start=5)
    #            a  <p> b  c </p> d <p> e </p> </p>
    assert pos.tolist() == [5, 6, 6, 7, 7, 8, 9, 9, 9, 9]


def test_synthetic_batches_follow_the_same_layout(tmp_path):
    """the bench's synthetic batches (data/synthetic.py) and prepare_vlm_inputs agree on prefix layout and position rule"""
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    st = SpecialTokens(base_vocab=32000)
    batch = make_batch([(3, 1, 32, 32)], [(1, 16, 16)], [(1, 2, 2)], [12], tok=st, seed=0, grounding=True, n_pairs=1, device='cpu')
    vi = batch['vlm_inputs']
    n_img = 1 + 2                                   # 2x2 patches pooled 2x2 -> 1, + boi + eoi
    assert vi['token_type_ids'][0, :2 + n_img].tolist() == [0] + [1] * n_img + [0]
    assert vi['position_ids'][0, :2 + n_img].tolist() == [0, 1] + [2] * (n_img - 2) + [3, 4]
    text = vi['input_ids'][0, 2 + n_img:]
    tok = FakeTokenizer()
    assert torch.equal(vi['position_ids'][0, 2 + n_img:], get_text_position_ids(text, tok, start=5))
