"""Run logger behind the reference's module-level call surface
(public names of rllab/misc/logger.py:113-348: ``log``, ``prefix``, ``record_tabular``,
``dump_tabular``, ``add_tabular_output`` ..., ``save_itr_params``).

The engine's own structure: one ``Logger`` object owning
  * a stack of text prefixes and a stack of tabular-key prefixes,
  * the pending table row (ordered key -> string),
  * a set of sinks -- ``TextSink`` (append-mode log files) and ``CsvSink`` (one CSV per
    experiment, header written with the first row) -- keyed by file name,
  * the snapshot policy (dir, mode all / last / gap / none, gap),
and, because the data-parallel run has one process per GPU that all execute the same
``train()`` loop, a *primary* flag: only the primary process (rank 0) prints, writes
sinks and snapshots; every rank still records its table row so ``get_tabular`` works
everywhere.  The module functions below are bound methods of the singleton ``_L``.

Tabular keys emitted by the hot path are the user-visible parity surface and keep the
reference's names.
"""
import csv
import datetime
import json
import os
import sys
from contextlib import contextmanager


class TextSink(object):
    """Append-mode log file, opened by the first line actually written (a non-primary rank that registers the sink
    before ``init_process_group`` therefore never touches the file)."""

    def __init__(self, path):
        self.path, self.fh = path, None

    def write_line(self, line):
        if self.fh is None:
            _ensure_parent(self.path)
            self.fh = open(self.path, "a")
        self.fh.write(line + "\n")
        self.fh.flush()

    def close(self):
        if self.fh is not None:
            self.fh.close()
            self.fh = None


class CsvSink(object):
    """One CSV per experiment; created (mode 'w') and given its header by the first row actually written."""

    def __init__(self, path):
        self.path, self.fh = path, None
        self.columns = None

    def write_row(self, row):
        if self.fh is None:
            _ensure_parent(self.path)
            self.fh = open(self.path, "w")
        if self.columns is None:
            self.columns = list(row)
            csv.writer(self.fh).writerow(self.columns)
        writer = csv.DictWriter(self.fh, fieldnames=self.columns, extrasaction="ignore", restval="")
        writer.writerow(row)
        self.fh.flush()

    def close(self):
        if self.fh is not None:
            self.fh.close()
            self.fh = None


def _ensure_parent(path):
    parent = os.path.dirname(os.path.abspath(path))
    os.makedirs(parent, exist_ok=True)


class Logger(object):
    SNAPSHOT_MODES = ("all", "last", "gap", "none")

    def __init__(self):
        self.text_prefixes, self.key_prefixes = [], []
        self.row = {}                       # pending table row, insertion ordered
        self.text_sinks, self.csv_sinks = {}, {}
        self.snapshot_dir, self.snapshot_mode, self.snapshot_gap = None, "all", 1
        self.tabular_only = False
        self.quiet = False
        self.primary = None                 # None: decide lazily from torch.distributed

    # -- who writes -------------------------------------------------------------------------------------------
    def is_primary(self):
        if self.primary is not None:
            return self.primary
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_rank() == 0
        except ImportError:
            pass
        # before the process group exists (run_experiment_lite opens the sinks before a user script calls
        # init_process_group): the launcher's environment already says which rank this process will be
        try:
            return int(os.environ.get("RANK", "0")) == 0
        except ValueError:
            return True

    def set_primary(self, flag):
        self.primary = flag

    def set_quiet(self, q=True):
        self.quiet = q

    # -- sinks ------------------------------------------------------------------------------------------------
    def add_text_output(self, file_name):
        # registered on every rank, opened lazily by the first PRIMARY write (is_primary is re-checked per write)
        if file_name not in self.text_sinks:
            self.text_sinks[file_name] = TextSink(file_name)

    def remove_text_output(self, file_name):
        sink = self.text_sinks.pop(file_name, None)
        if sink is not None:
            sink.close()

    def add_tabular_output(self, file_name):
        if file_name not in self.csv_sinks:
            self.csv_sinks[file_name] = CsvSink(file_name)

    def remove_tabular_output(self, file_name):
        sink = self.csv_sinks.pop(file_name, None)
        if sink is not None:
            sink.close()

    # -- text -------------------------------------------------------------------------------------------------
    def push_prefix(self, p):
        self.text_prefixes.append(p)

    def pop_prefix(self):
        self.text_prefixes.pop()

    @contextmanager
    def prefix(self, key):
        self.push_prefix(key)
        try:
            yield
        finally:
            self.pop_prefix()

    def log(self, s, with_prefix=True, with_timestamp=True, color=None):
        if not self.is_primary():
            return
        if (self.tabular_only or self.quiet) and not self.text_sinks:
            return                        # nobody reads it: no timestamp, no string (these sit between kernel launches)
        line = ("".join(self.text_prefixes) if with_prefix else "") + s
        if with_timestamp:
            line = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f") + " | " + line
        if not (self.tabular_only or self.quiet):
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
        for sink in self.text_sinks.values():
            sink.write_line(line)

    # -- table ------------------------------------------------------------------------------------------------
    def push_tabular_prefix(self, key):
        self.key_prefixes.append(key)

    def pop_tabular_prefix(self):
        self.key_prefixes.pop()

    @contextmanager
    def tabular_prefix(self, key):
        self.push_tabular_prefix(key)
        try:
            yield
        finally:
            self.pop_tabular_prefix()

    def record_tabular(self, key, val):
        self.row["".join(self.key_prefixes) + str(key)] = str(val)

    def get_tabular(self):
        """Current (not yet dumped) table row as a dict key -> string value."""
        return dict(self.row)

    def render_row(self):
        kw = max(len(k) for k in self.row)
        vw = max(len(v) for v in self.row.values())
        rule = "-" * (kw + vw + 5)
        return [rule] + ["%s  %s" % (k.ljust(kw), v.rjust(vw)) for k, v in self.row.items()] + [rule]

    def dump_tabular(self, *args, **kwargs):
        if not self.row:
            return
        if self.is_primary():
            if not self.quiet:
                lines = self.render_row()
                if self.tabular_only:
                    sys.stdout.write("\n".join(lines) + "\n")
                else:
                    for line in lines:
                        self.log(line, *args, **kwargs)
            for sink in self.csv_sinks.values():
                sink.write_row(self.row)
        self.row = {}

    def record_tabular_misc_stat(self, key, values, placement="back"):
        """Average / Std / Median / Min / Max of ``values`` under ``key`` (suffix, or prefix with placement='front');
        NaNs for an empty list (rllab/misc/logger.py:330-348)."""
        import numpy as np
        for stat, fn in (("Average", np.average), ("Std", np.std), ("Median", np.median), ("Min", np.min),
                         ("Max", np.max)):
            name = stat + key if placement == "front" else key + stat
            self.record_tabular(name, fn(values) if len(values) > 0 else np.nan)

    # -- snapshots --------------------------------------------------------------------------------------------
    def snapshot_path(self, itr):
        """File the snapshot of iteration ``itr`` goes to under the current mode, or None (reference modes,
        rllab/misc/logger.py:216-232)."""
        mode = self.snapshot_mode
        if mode not in self.SNAPSHOT_MODES:
            raise NotImplementedError("snapshot mode %r" % (mode,))
        if mode == "none" or (mode == "gap" and itr % self.snapshot_gap != 0):
            return None
        name = "params.pkl" if mode == "last" else "itr_%d.pkl" % itr
        return os.path.join(self.snapshot_dir, name)

    def save_itr_params(self, itr, params):
        if not self.snapshot_dir or not self.is_primary():
            return                         # plain example scripts set no snapshot dir
        path = self.snapshot_path(itr)
        if path is not None:
            import joblib
            joblib.dump(params, path, compress=3)


_L = Logger()

log = _L.log
prefix = _L.prefix
push_prefix = _L.push_prefix
pop_prefix = _L.pop_prefix
tabular_prefix = _L.tabular_prefix
push_tabular_prefix = _L.push_tabular_prefix
pop_tabular_prefix = _L.pop_tabular_prefix
record_tabular = _L.record_tabular
record_tabular_misc_stat = _L.record_tabular_misc_stat
get_tabular = _L.get_tabular
dump_tabular = _L.dump_tabular
add_text_output = _L.add_text_output
remove_text_output = _L.remove_text_output
add_tabular_output = _L.add_tabular_output
remove_tabular_output = _L.remove_tabular_output
save_itr_params = _L.save_itr_params
set_quiet = _L.set_quiet
set_primary = _L.set_primary
is_primary = _L.is_primary


def set_snapshot_dir(dir_name):
    _L.snapshot_dir = dir_name


def get_snapshot_dir():
    return _L.snapshot_dir


def set_snapshot_mode(mode):
    _L.snapshot_mode = mode


def get_snapshot_mode():
    return _L.snapshot_mode


def set_snapshot_gap(gap):
    _L.snapshot_gap = gap


def get_snapshot_gap():
    return _L.snapshot_gap


def set_log_tabular_only(flag):
    _L.tabular_only = flag


def get_log_tabular_only():
    return _L.tabular_only


def _dump_json(log_file, payload):
    _ensure_parent(log_file)
    with open(log_file, "w") as fh:
        json.dump(payload, fh, indent=2, sort_keys=True, default=repr)


def log_variant(log_file, variant_data):
    """variant.json of an experiment (rllab/misc/logger.py:321-327); values JSON cannot carry go in as ``repr``."""
    if hasattr(variant_data, "dump"):
        variant_data = variant_data.dump()
    _dump_json(log_file, variant_data)


def log_parameters_lite(log_file, args):
    """params.json from an argparse namespace (rllab/misc/logger.py:301-318, without the stub decoding that only
    the reference's subprocess launcher needs)."""
    _dump_json(log_file, dict(vars(args)))
