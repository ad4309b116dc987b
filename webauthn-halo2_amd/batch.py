"""Batch driver for independent proofs (BASELINE.json configs[3]; SURVEY.md §8d "config 4", §8e).

The reference's concurrency model is one Rocket worker thread per request, each running its own
`create_proof` (proving-server/src/main.rs:457-472).  Here that is: one `Pipeline` (= one `zk_ctx`: HIP
device + streams + resident SRS / proving key + host thread) per in-flight proof, one or more pipelines per
GPU, jobs dealt round-robin; a job is one request — here one synthetic witness of seed 0x5eed0019 + i
 — and its result is the proof bytes.  Replicas only: no data-path collective, nothing is
exchanged between pipelines (§8e).

    assign(jobs, rank, world)            round-robin shard of a job list
    synthesize_jobs(params, jobs)        host witness generation for a list of jobs (process pool)
    Pipeline(device, params)             resident SRS + key on one device; .load(job, cols) / .prove(job)
    run(pipelines, jobs, transcript)     drains `jobs` over the pipelines (one host thread each) -> {job: proof}
"""
import os
import threading

import numpy as np

from . import circuit
from .engine import ZK_TRANSCRIPT_BLAKE2B, Engine

BASE_SEED = 0x5EED0019


def job_seed(i: int) -> int:
    return BASE_SEED + i


def job_rng_seed(i: int) -> bytes:
    """A FIXED create_proof RNG stream for job i: makes a batch reproducible and comparable with lone proofs (tests,
    bench.py).  The blinding of such a proof is predictable from the public job index, i.e. the proof is NOT
    zero-knowledge — production callers leave Pipeline's default (os.urandom, as the reference draws from OsRng,
    ecdsa_p256.rs:412)."""
    return (0x9E3779B97F4A7C15 * (i + 1) % (1 << 256)).to_bytes(32, "little")


def assign(jobs, rank: int, world: int):
    """Round-robin shard of the job list for `rank` of `world` (weak scaling: no exchange)."""
    return list(jobs)[rank::world]


def _synth_one(args):
    params, i, worst = args
    asg = circuit.synthesize(params, job_seed(i), worst_case=worst)
    return i, [asg.to_limbs(col) for col in asg.advice]


def synthesize_jobs(params, jobs, processes=None, worst_case=False):
    """Advice columns (canonical limbs) of every job: {job: [(n, 4) uint64 per advice column]}.
    Witness generation is host work (in the reference: ECDSACircuit::synthesize, ecdsa_p256.rs:117-206);
    jobs are independent, so a process pool spreads them over the host cores."""
    jobs = list(jobs)
    if not jobs:
        return {}
    if processes is None:
        import os
        processes = min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 32)
    args = [(params, i, worst_case) for i in jobs]
    if processes <= 1 or len(jobs) == 1:
        return dict(_synth_one(a) for a in args)
    import multiprocessing as mp
    with mp.get_context("fork").Pool(processes) as pool:
        return dict(pool.imap_unordered(_synth_one, args, chunksize=1))


def structure(params):
    """The witness-independent half of synthesis: fixed columns + copy constraints (one proving key for all jobs)."""
    asg = circuit.synthesize(params, 0)
    return np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies


def _quietly(fn, *args):
    """Run a clean-up step; a failure is reported on stderr and returned, never raised (the caller is already unwinding)."""
    try:
        fn(*args)
        return None
    except Exception as e:  # noqa: BLE001 — whatever the engine raises
        import sys
        print("webauthn-halo2_amd.batch: clean-up step %s failed: %s" % (getattr(fn, "__name__", fn), e), file=sys.stderr)
        return e


class Pipeline:
    """One proof in flight: a zk_ctx on `device` with the SRS of params.degree and the proving key resident."""

    def __init__(self, device, params, fixed=None, copies=None, engine_factory=Engine, deterministic_seeds=False,
                 share_srs_with=None):
        """deterministic_seeds: blinding from job_rng_seed(job) instead of the OS entropy source — reproducible
        proofs for tests and benchmarks, at the price of zero-knowledge (see job_rng_seed).
        share_srs_with: another Pipeline on the same device whose resident SRS and window tables this one uses too
        (zk_ctx_create_shared) — the further pipelines of a GPU need no copy of their own."""
        self.params = params
        self.device = device
        self.deterministic_seeds = deterministic_seeds
        if share_srs_with is not None:
            self.eng = Engine(device, share_with=share_srs_with.eng)
            configure = getattr(engine_factory, "configure", None)  # a factory's per-engine settings also reach the sharing pipelines
            if configure is not None:
                configure(self.eng)
        else:
            self.eng = engine_factory(device)
            self.eng.srs_setup(params.degree)
        if fixed is None:
            fixed, copies = structure(params)
        self.pk = self.eng.keygen(params, fixed, copies)
        self.resident = {}  # job -> [Poly]
        self._spare = []    # buffers of finished jobs, reused by the next load

    def load(self, job, columns):
        """Ship a job's advice columns to the device (the per-request H2D of a real host)."""
        n = 1 << self.params.degree
        polys = []
        for col in columns:
            h = self._spare.pop() if self._spare else self.eng.poly(n)  # a finished job's buffer, if there is one
            self.eng.upload_canonical(h, col)
            polys.append(h)
        self.resident[job] = polys

    def reload(self, job, columns):
        """The per-request H2D into the job's EXISTING device buffers (a host that keeps its request slots: no hipMalloc /
        hipFree — the latter waits for the whole device — between proofs); falls back to `load` for a job not yet resident."""
        polys = self.resident.get(job)
        if polys is None or len(polys) != len(columns):
            self.unload(job)
            return self.load(job, columns)
        for h, col in zip(polys, columns):
            self.eng.upload_canonical(h, col)

    def loader(self):
        """A loader context beside this pipeline (shared SRS, own stream): `stage` uploads a job's columns on it — from another
        host thread, while this pipeline proves — and `adopt` hands them over between two proofs."""
        if getattr(self, "_loader", None) is None:
            self._loader = Engine(self.device, share_with=self.eng)
        return self._loader

    def stage(self, columns):
        """Upload + convert a job's advice columns on the loader context and detach them (thread-safe against `prove`: the
        loader context has its own lock and stream) -> staged columns for `adopt`."""
        ld = self.loader()
        n = 1 << self.params.degree
        staged = []
        h = None
        try:
            for col in columns:
                h = ld.poly(n)
                ld.upload_canonical(h, col)
                staged.append(ld.poly_detach(h))
                h = None
        except Exception:
            # a later column failed: nobody will ever attach the ones already detached — give their memory back.  The clean-up
            # itself may fail (after a device fault every call does): that must neither replace the original error nor stop
            # the remaining tokens from being discarded
            if h is not None and h.h:
                _quietly(h.free)
            for d in staged:
                _quietly(ld.poly_discard, d)
            raise
        return staged

    def discard(self, staged):
        """Drop staged columns that will not be adopted (the request was cancelled, the pipeline is closing).  Every token is
        tried; the first failure is raised once all have been."""
        first = None
        for d in staged:
            err = _quietly(self.loader().poly_discard, d)
            first = first or err
        if first is not None:
            raise first

    def adopt(self, job, staged):
        """Make staged columns this pipeline's resident advice of `job` (no copy)."""
        self.unload(job)
        self.resident[job] = [self.eng.poly_attach(d) for d in staged]

    def prove(self, job, transcript=ZK_TRANSCRIPT_BLAKE2B, keep=False, rng_seed=None):
        if rng_seed is None:
            rng_seed = job_rng_seed(job) if self.deterministic_seeds else os.urandom(32)
        proof = self.eng.prove(self.pk, self.resident[job], rng_seed, transcript)
        if not keep:
            self.unload(job)
        return proof

    def prove_lockstep(self, jobs, transcript=ZK_TRANSCRIPT_BLAKE2B, keep=False, rng_seeds=None):
        """The proofs of `jobs` (all resident on this pipeline) in ONE lock-step batch (zk_prove_batch): the same commitment of
        every job shares an MSM pass, the same transform an NTT launch.  Returns the proofs in the order of `jobs`; each is
        byte-identical to prove(job)."""
        jobs = list(jobs)
        if rng_seeds is None:
            rng_seeds = [job_rng_seed(j) if self.deterministic_seeds else os.urandom(32) for j in jobs]
        proofs = self.eng.prove_batch(self.pk, [self.resident[j] for j in jobs], rng_seeds, transcript)
        if not keep:
            for j in jobs:
                self.unload(j)
        return proofs

    def unload(self, job):
        """The job's buffers go to the pipeline's spare list, not back to the device allocator: hipFree waits for the whole
        device — for the other pipelines' kernels too — so a drain that freed per job would stall everybody per proof."""
        polys = self.resident.pop(job, [])
        room = max(0, 4 * max(1, len(polys)) - len(self._spare))  # a few request slots, not an ever-growing list
        self._spare.extend(polys[:room])
        for p in polys[room:]:
            p.free()

    def close(self):
        for job in list(self.resident):
            self.unload(job)
        for p in self._spare:
            p.free()
        self._spare = []
        if getattr(self, "_loader", None) is not None:
            self._loader.close()
        self.eng.close()


def run_lockstep(pipelines, jobs, lockstep, transcript=ZK_TRANSCRIPT_BLAKE2B, keep=False):
    """Drain `jobs` (pipeline q holds jobs[q::len(pipelines)], as for `run`) in lock-step batches of `lockstep` proofs per
    pipeline: one host thread per pipeline, each proving its share `lockstep` jobs at a time (the last batch may be
    shorter).  Returns {job: proof bytes} — the same bytes as `run`."""
    jobs = list(jobs)
    out = {}
    errs = []

    def work(q):
        try:
            mine = jobs[q::len(pipelines)]
            for i in range(0, len(mine), lockstep):
                group = mine[i:i + lockstep]
                for j, pf in zip(group, pipelines[q].prove_lockstep(group, transcript, keep)):
                    out[j] = pf
        except Exception as e:  # surfaced to the caller below
            errs.append(e)

    if len(pipelines) == 1:
        work(0)
    else:
        ths = [threading.Thread(target=work, args=(q,)) for q in range(len(pipelines))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    if errs:
        raise errs[0]
    return out


def run(pipelines, jobs, transcript=ZK_TRANSCRIPT_BLAKE2B, keep=False, on_done=None):
    """Drain `jobs` (already loaded: pipeline q holds jobs[q::len(pipelines)]) — one host thread per pipeline,
    as the reference has one worker thread per request.  Returns {job: proof bytes}."""
    jobs = list(jobs)
    out = {}
    errs = []

    def work(q):
        try:
            for j in jobs[q::len(pipelines)]:
                out[j] = pipelines[q].prove(j, transcript, keep)
                if on_done:
                    on_done(j)
        except Exception as e:  # surfaced to the caller below
            errs.append(e)

    if len(pipelines) == 1:
        work(0)
    else:
        ths = [threading.Thread(target=work, args=(q,)) for q in range(len(pipelines))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    if errs:
        raise errs[0]
    return out
