"""Collects the parity numbers the -m gpu tests measure (worst loss / output / gradient errors per fixture and storage type)
and writes them to one text file at the end of the session: $VKAS_PARITY_REPORT, default gpurun_out/parity_report.txt.
The copy judged with a round is committed as profiles/parity_rNN.txt."""
import os
import time

_ROWS = []


def record(test: str, quantity: str, value: float, bound=None, note: str = ''):
    _ROWS.append((test, quantity, float(value), bound, note))


def write_report():
    if not _ROWS:
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get('VKAS_PARITY_REPORT') or os.path.join(root, 'gpurun_out', 'parity_report.txt')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    w = max(len(r[0]) for r in _ROWS)
    wq = max(len(r[1]) for r in _ROWS)
    with open(path, 'w') as f:
        f.write('# parity numbers measured by `python -m pytest tests -m gpu` (tests/parity_log.py), %s\n'
                % time.strftime('%Y-%m-%d %H:%M:%S'))
        f.write('# errors are norm-wise relative (||a - ref|| / ||ref||, fp64) unless the quantity says otherwise;\n')
        f.write('# "bound" is the value the test asserts\n')
        f.write('%-*s  %-*s  %12s  %10s  %s\n' % (w, 'test', wq, 'quantity', 'measured', 'bound', 'note'))
        for t, q, v, b, n in _ROWS:
            f.write('%-*s  %-*s  %12.4e  %10s  %s\n' % (w, t, wq, q, v, ('%.1e' % b) if b is not None else '-', n))
    return path
