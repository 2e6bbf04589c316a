"""Prefix / tabular logger with the reference's call surface
(rllab/misc/logger.py:113-232): ``log``, ``prefix``, ``record_tabular``,
``dump_tabular``, ``add_tabular_output`` (CSV), ``save_itr_params`` (joblib
snapshots by mode all/last/gap/none).  Tabular keys emitted by the hot path are
the user-visible parity surface and keep the reference's names."""
import csv
import datetime
import os
import sys
from contextlib import contextmanager

_prefixes = []
_prefix_str = ''
_tabular_prefixes = []
_tabular_prefix_str = ''
_tabular = []
_text_outputs = []
_tabular_outputs = []
_text_fds = {}
_tabular_fds = {}
_tabular_header_written = set()
_snapshot_dir = None
_snapshot_mode = 'all'
_snapshot_gap = 1
_log_tabular_only = False
_quiet = False


def set_quiet(q=True):
    global _quiet
    _quiet = q


def _add_output(file_name, arr, fds, mode='a'):
    if file_name not in arr:
        os.makedirs(os.path.dirname(os.path.abspath(file_name)), exist_ok=True)
        arr.append(file_name)
        fds[file_name] = open(file_name, mode)


def _remove_output(file_name, arr, fds):
    if file_name in arr:
        fds[file_name].close()
        del fds[file_name]
        arr.remove(file_name)


def push_prefix(prefix):
    global _prefix_str
    _prefixes.append(prefix)
    _prefix_str = ''.join(_prefixes)


def pop_prefix():
    global _prefix_str
    del _prefixes[-1]
    _prefix_str = ''.join(_prefixes)


def add_text_output(file_name):
    _add_output(file_name, _text_outputs, _text_fds, mode='a')


def remove_text_output(file_name):
    _remove_output(file_name, _text_outputs, _text_fds)


def add_tabular_output(file_name):
    _add_output(file_name, _tabular_outputs, _tabular_fds, mode='w')


def remove_tabular_output(file_name):
    if file_name in _tabular_outputs and _tabular_fds[file_name] in _tabular_header_written:
        _tabular_header_written.remove(_tabular_fds[file_name])
    _remove_output(file_name, _tabular_outputs, _tabular_fds)


def set_snapshot_dir(dir_name):
    global _snapshot_dir
    _snapshot_dir = dir_name


def get_snapshot_dir():
    return _snapshot_dir


def get_snapshot_mode():
    return _snapshot_mode


def set_snapshot_mode(mode):
    global _snapshot_mode
    _snapshot_mode = mode


def get_snapshot_gap():
    return _snapshot_gap


def set_snapshot_gap(gap):
    global _snapshot_gap
    _snapshot_gap = gap


def set_log_tabular_only(log_tabular_only):
    global _log_tabular_only
    _log_tabular_only = log_tabular_only


def get_log_tabular_only():
    return _log_tabular_only


def log(s, with_prefix=True, with_timestamp=True, color=None):
    out = s
    if with_prefix:
        out = _prefix_str + out
    if with_timestamp:
        now = datetime.datetime.now()
        out = "%s | %s" % (now.strftime('%Y-%m-%d %H:%M:%S.%f'), out)
    if not _log_tabular_only and not _quiet:
        print(out)
        sys.stdout.flush()
    for fd in list(_text_fds.values()):
        fd.write(out + '\n')
        fd.flush()


def record_tabular(key, val):
    _tabular.append((_tabular_prefix_str + str(key), str(val)))


def push_tabular_prefix(key):
    global _tabular_prefix_str
    _tabular_prefixes.append(key)
    _tabular_prefix_str = ''.join(_tabular_prefixes)


def pop_tabular_prefix():
    global _tabular_prefix_str
    del _tabular_prefixes[-1]
    _tabular_prefix_str = ''.join(_tabular_prefixes)


@contextmanager
def prefix(key):
    push_prefix(key)
    try:
        yield
    finally:
        pop_prefix()


@contextmanager
def tabular_prefix(key):
    push_tabular_prefix(key)
    yield
    pop_tabular_prefix()


def get_tabular():
    """Current (not yet dumped) tabular rows as a dict key -> string value."""
    return dict(_tabular)


def dump_tabular(*args, **kwargs):
    if len(_tabular) > 0:
        if not _quiet:
            width = max(len(k) for k, _ in _tabular)
            vwidth = max(len(v) for _, v in _tabular)
            bar = '-' * (width + vwidth + 5)
            lines = [bar] + ["%s  %s" % (k.ljust(width), v.rjust(vwidth)) for k, v in _tabular] + [bar]
            if _log_tabular_only:
                print('\n'.join(lines))
            else:
                for line in lines:
                    log(line, *args, **kwargs)
        tabular_dict = dict(_tabular)
        for tabular_fd in list(_tabular_fds.values()):
            writer = csv.DictWriter(tabular_fd, fieldnames=list(tabular_dict.keys()))
            if tabular_fd not in _tabular_header_written:
                writer.writeheader()
                _tabular_header_written.add(tabular_fd)
            writer.writerow(tabular_dict)
            tabular_fd.flush()
        del _tabular[:]


def save_itr_params(itr, params):
    """joblib snapshots, reference modes (logger.py:216-232); no-op without a
    snapshot dir (plain example scripts set none)."""
    if not _snapshot_dir:
        return
    import joblib
    if _snapshot_mode == 'all':
        joblib.dump(params, os.path.join(_snapshot_dir, 'itr_%d.pkl' % itr), compress=3)
    elif _snapshot_mode == 'last':
        joblib.dump(params, os.path.join(_snapshot_dir, 'params.pkl'), compress=3)
    elif _snapshot_mode == 'gap':
        if itr % _snapshot_gap == 0:
            joblib.dump(params, os.path.join(_snapshot_dir, 'itr_%d.pkl' % itr), compress=3)
    elif _snapshot_mode == 'none':
        pass
    else:
        raise NotImplementedError


def record_tabular_misc_stat(key, values, placement='back'):
    """Average / Std / Median / Min / Max of ``values`` under ``key`` (prefix, or suffix with placement='front');
    NaNs for an empty list (rllab/misc/logger.py:330-348)."""
    import numpy as np
    label = (lambda stat: stat + key) if placement == 'front' else (lambda stat: key + stat)
    stats = (("Average", np.average), ("Std", np.std), ("Median", np.median), ("Min", np.min), ("Max", np.max))
    for name, fn in stats:
        record_tabular(label(name), fn(values) if len(values) > 0 else np.nan)


def log_variant(log_file, variant_data):
    """variant.json of an experiment (rllab/misc/logger.py:321-327); values that JSON cannot carry are written as
    their ``repr``."""
    import json
    if hasattr(variant_data, "dump"):
        variant_data = variant_data.dump()
    d = os.path.dirname(log_file)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(log_file, "w") as fh:
        json.dump(variant_data, fh, indent=2, sort_keys=True, default=repr)


def log_parameters_lite(log_file, args):
    """params.json from an argparse namespace (rllab/misc/logger.py:301-318, without the stub decoding that only
    the reference's subprocess launcher needs)."""
    import json
    d = os.path.dirname(log_file)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(log_file, "w") as fh:
        json.dump(dict(vars(args)), fh, indent=2, sort_keys=True, default=repr)
