"""zkoracle — CPU restatement of the reference's proving hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`webauthn-halo2_amd/`,
`libzkmi355.so`) may import, link or execute anything under `oracle/`; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
do, and there only as the checker.

The algorithms restated here live in third-party crates that are NOT under
/root/reference (halo2_proofs PSE fork, halo2curves, snark-verifier; pulled by
git branch with no lockfile — reference halo2-circuits/Cargo.toml:12-15,
.gitignore:10).  They are restated from the published algorithms and anchored
to the reference's own artefacts (SURVEY.md §8c K1–K7):

  K1  tau / [tau]G2       proving-server/P256Verifier.yul:1125-1134
  K2  table-column commit proving-server/P256Verifier.yul:889-890
  K3  k=17 verifying key  proving-server/P256Verifier.yul:34,880-980
  K4  domain constants    proving-server/P256Verifier.yul:17-18,289-323,767-775
  K5  golden EVM proof    contracts/test/P256Account.t.sol:120-121
  K6  proof sizes         halo2-circuits/src/results/ecdsa_bench.csv:2-10

Parity status: the Keccak/GWC (EVM) path is PINNED by K1–K5.  The
Blake2b/SHPLONK path is *parity unpinned* (the reference ships no bytes for
it, only sizes — K6).
"""
