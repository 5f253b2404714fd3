"""Host side of the generation path: `prepare_inputs_for_generation` (reference mmmm.py:368-406) replayed on
the token / position sequences the REFERENCE produced (fixture f12, oracle/make_golden.py::f12_generation). Integer work:
bit-exact. No kernel is touched, so this runs without a GPU."""
from pathlib import Path
import types

import torch

F12 = torch.load(Path(__file__).parent / 'golden' / 'f12_generation.pt', weights_only=False)


def test_position_rule_of_prepare_inputs_for_generation():
    from mmmm_amd.models.mmmm import MMMMForCausalLM
    me = types.SimpleNamespace(tokenizer=types.SimpleNamespace(bop_token_id=F12['bop_token_id'], eop_token_id=F12['eop_token_id']))
    for b, rec in enumerate(F12['samples']):
        ids = rec['prefill']['input_ids'].clone()
        tt = rec['prefill']['token_type_ids'].clone()
        pos = rec['prefill']['position_ids'].clone()
        for t, st in enumerate(rec['steps']):
            ids = torch.cat([ids, torch.tensor([[st['token']]])], 1)
            tt = torch.cat([tt, torch.zeros(1, 1, dtype=torch.long)], 1)
            pos = torch.cat([pos, pos[:, -1:] + 1], 1)
            inputs = MMMMForCausalLM.prepare_inputs_for_generation(
                me, ids, token_type_ids=tt, position_ids=pos, past_key_values=[object()], attention_mask=torch.ones_like(ids),
                patch_size=None, pool_size=None, use_cache=True)
            assert inputs['input_ids'].shape == (1, 1) and int(inputs['input_ids']) == st['token']
            assert int(inputs['position_ids']) == st['position_id'] and inputs['token_type_ids'].shape == (1, 1)
            assert inputs['use_cache'] is True
        assert torch.equal(pos, rec['final_position_ids'])          # corrected in place, as the reference does
    # without a cache the prompt passes through untouched
    pre = F12['samples'][0]['prefill']
    inputs = MMMMForCausalLM.prepare_inputs_for_generation(me, pre['input_ids'], token_type_ids=pre['token_type_ids'],
                                                           position_ids=pre['position_ids'], past_key_values=None,
                                                           attention_mask=pre['attention_mask'], patch_size=[(1, 8, 8)], pool_size=[(1, 2, 2)])
    assert inputs['input_ids'] is pre['input_ids'] and inputs['position_ids'] is pre['position_ids']


def test_generate_output_sequences_layout():
    from mmmm_amd.models.mmmm import GenerateOutput
    ids = torch.tensor([[5, 6, 7, 0], [8, 9, 0, 0]])
    out = GenerateOutput(new_tokens=torch.tensor([[1, 2], [3, 4]]), new_position_ids=torch.zeros(2, 2, dtype=torch.long),
                         prompt_lengths=torch.tensor([3, 2]))
    assert out.sequences(ids).tolist() == [[5, 6, 7, 1, 2, 0], [8, 9, 3, 4, 0, 0]]
