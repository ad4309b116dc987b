"""Batch driver for independent proofs (BASELINE.json configs[3]): jobs are assigned round-robin,
one proof in flight per GPU, no collective.  Mirrors the request shape of the reference's
/prove endpoints (proving-server/src/main.rs:39-79): a job is one (r, s, pubkey, msghash) tuple; here
it selects the synthetic witness seed (0x5eed0019 + i)."""

BASE_SEED = 0x5EED0019


def job_seed(i: int) -> int:
    return BASE_SEED + i


def assign(jobs, rank: int, world: int):
    """Round-robin shard of the job list for `rank` of `world` (weak scaling: no exchange)."""
    return list(jobs)[rank::world]


def run(engine, pk, params, jobs, transcript, make_witness, upload):
    """Prove every job in `jobs` on `engine`; returns {job: proof bytes}."""
    out = {}
    for i in jobs:
        asg = make_witness(params, job_seed(i))
        polys = upload(engine, asg)
        out[i] = engine.prove(pk, polys, i.to_bytes(32, "little"), transcript)
        for p in polys:
            p.free()
    return out
