"""ezpz_amd: the MI355X-native Newton / Levenberg-Marquardt constraint-solve path of KittyCAD/ezpz.

A C-ABI shared library (include/ezpz_amd.h; HIP kernels for gfx950 + C++ host symbolic phase) and this
thin host mirror of the `ezpz` crate's solve API.  There is no CPU fallback: importing works anywhere the
library builds, solving needs a HIP device.
"""
from . import textual
from ._lib import CONSTRAINT_DTYPE, STATUS_DTYPE, lib
from .api import (Angle, AngleKind, CircleSide, Config, Constraint, ConstraintRequest, DatumCircle, DatumCircularArc,
                  DatumDistance, DatumLineSegment, DatumPoint, FailureOutcome, FreedomAnalysis, IdGenerator, LineSide,
                  MixedBatch, MultiSystem, NonLinearSystemError, RawResult, SolveOutcome, SolveOutcomeFreedomAnalysis, System, TEAM_AUTO_LATENCY, TEAM_AUTO_LISTS, TEAM_BATCH_LANES, TEAM_FRONTS, TEAM_LATENCY_PHASES, TEAM_LATENCY_RECORDS, TEAM_LATENCY_WAVE, Warning, WarningContent, analyze, device_count, launch_policy, solve,
                  host_register, host_unregister, resolve_sides, solve_analysis, specialized_source, solve_batch, solve_batch_mixed, solve_batch_mixed_multi, solve_batch_multi, solve_records, stack_records)

__all__ = [
    "Angle", "AngleKind", "CircleSide", "Config", "Constraint", "ConstraintRequest", "DatumCircle", "DatumCircularArc",
    "DatumDistance", "DatumLineSegment", "DatumPoint", "FailureOutcome", "FreedomAnalysis", "IdGenerator", "LineSide",
    "MixedBatch", "MultiSystem", "NonLinearSystemError", "RawResult", "SolveOutcome", "SolveOutcomeFreedomAnalysis", "System", "TEAM_AUTO_LATENCY", "TEAM_AUTO_LISTS", "TEAM_BATCH_LANES", "TEAM_FRONTS", "TEAM_LATENCY_PHASES", "TEAM_LATENCY_RECORDS", "TEAM_LATENCY_WAVE", "Warning", "WarningContent", "analyze",
    "device_count", "launch_policy", "host_register", "host_unregister", "resolve_sides", "solve", "specialized_source",
    "solve_analysis", "solve_batch", "solve_batch_mixed", "solve_batch_mixed_multi", "solve_batch_multi", "solve_records", "stack_records", "textual", "lib", "CONSTRAINT_DTYPE", "STATUS_DTYPE",
]
