"""MATLAB-syntax dump of the per-iteration trace records (jrlqp_trace_rec) in the format of the reference's Logger
(include/jrl-qp/utils/Logger.h:98-165): one `name(it+1).var = value;` line per logged value, `name(it+1).it = it;` first
(Logger::startIter, src/utils/Logger.cpp:23-27), the values of LogFlags::ACTIVE_SET / ITERATION_BASIC_DETAILS that the GPU
path records: the selected constraint and its status (DualSolver.cpp:116), the step lengths, the blocking position l and
whether the pass activates (DualSolver.cpp:138,153,159).  The device never formats text: this runs on the host on the
trace buffer of a finished solve (BatchSolver(..., trace_cap=N).trace_numpy(), Oracle.solve(..., trace_cap=N).trace)."""
from __future__ import annotations

import io


def _num(v):
    if isinstance(v, float):
        return repr(v) if v == v and abs(v) != float("inf") else ("NaN" if v != v else ("Inf" if v > 0 else "-Inf"))
    return str(int(v))


def _vec(v):
    return "[" + ", ".join(_num(float(e)) for e in v) + "]'"


def to_matlab(records, name="log", stream=None, vectors=None):
    """records: iterable of (p, status, l, kind, t1, t2) tuples or numpy structured records; returns the text (and writes
    it to `stream` if given).  kind: 0 full step + add, 1 partial step + drop, 2 dual-only step + drop.
    vectors: optional per-pass dicts with x, u, f, z, r (capi.DenseSolver.trace_vectors) -- the values the reference logs under
    LogFlags::ITERATION_BASIC_DETAILS (DualSolver.cpp:108: x, u, f_; :128-129: z, r, t), written in the reference's order."""
    out = io.StringIO()
    for it, rec in enumerate(records):
        p, status, l, kind, t1, t2 = (rec[k] for k in ("p", "status", "l", "kind", "t1", "t2")) if hasattr(rec, "dtype") else rec
        pre = f"{name}({it + 1})"
        t = min(float(t1), float(t2))
        out.write(f"{pre}.it = {it};\n")
        vec = vectors[it] if vectors is not None and it < len(vectors) else None
        if vec is not None:
            out.write(f"{pre}.x = {_vec(vec['x'])};\n")
            out.write(f"{pre}.u = {_vec(vec['u'])};\n")
            out.write(f"{pre}.f = {_num(float(vec['f']))};\n")
        out.write(f"{pre}.selectedConstraint = {_num(p)};\n")
        out.write(f"{pre}.status = {_num(status)};\n")
        out.write(f"{pre}.t1 = {_num(float(t1))};\n")
        out.write(f"{pre}.t2 = {_num(float(t2))};\n")
        if vec is not None:
            out.write(f"{pre}.z = {_vec(vec['z'])};\n")
            out.write(f"{pre}.r = {_vec(vec['r'])};\n")
        out.write(f"{pre}.t = {_num(t)};\n")
        out.write(f"{pre}.Activate = {'true' if int(kind) == 0 else 'false'};\n")
        out.write(f"{pre}.l = {_num(l)};\n")
    text = out.getvalue()
    if stream is not None:
        stream.write(text)
    return text
