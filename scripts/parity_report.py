#!/usr/bin/env python3
"""GPU box: per-tensor parity table HIP vs CPU oracle (outputs, intermediates, losses, gradients)."""
import sys
import traceback

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import parity_util as pu  # noqa: E402


def main():
    drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    case = pu.make_case()
    try:
        rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop)
        print(pu.format_report(rows))
        print('span indices equal:', idx_equal, h['start_index'].tolist(), o['start_index'].tolist(),
              h['end_index'].tolist(), o['end_index'].tolist())
    except Exception:
        traceback.print_exc()


if __name__ == '__main__':
    main()
