"""oracle/ -- CPU checkers.  TEST INFRASTRUCTURE ONLY.

Nothing under this package is imported, linked or executed by the product
(``rllab_amd``).  Its only users are ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.
"""
