"""UPPER-BOUND experiment (not a valid benchmark): how much of the step is the ~560 tiny ATen launches of the per-target Hungarian
cost matrices? Runs bench.main() with `_match_costs` memoised per (sample shapes) after its first evaluation — the costs go stale,
the number is meaningless as throughput; only the DIFFERENCE to a plain run says what fusing those ops could buy.
usage: python tools/ab_match_cost_memo.py [bench.py arguments]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss  # noqa: E402

_orig = InstanceSamLoss._match_costs
_memo = {}


def memo(self, boxes_reg, disc_logit, boxes_label, offs):
    key = (tuple(boxes_reg.shape), tuple(boxes_label.shape), tuple(offs))
    if key not in _memo:
        _memo[key] = _orig(self, boxes_reg, disc_logit, boxes_label, offs)
    return _memo[key]


InstanceSamLoss._match_costs = memo
bench.main()
