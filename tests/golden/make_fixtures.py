"""Generates tests/golden/*.json|hex from the reference's own artefacts.  Run in the
build container only (needs /root/reference); the outputs are committed data:

  golden_proof_k17_evm.hex  the 2720-byte EVM/GWC proof embedded in
                            contracts/test/P256Account.t.sol:120-121
  vk_k17.json               the k=17 verifying key constants that snark-verifier baked
                            into proving-server/P256Verifier.yul (transcript_repr at :34,
                            12 commitments at :880-980) and the Fiat-Shamir challenges
                            obtained by running that Yul on the golden proof with
                            oracle/tools/yul_exec.py
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "tools")]
REF = "/root/reference"

sol = open(f"{REF}/contracts/test/P256Account.t.sol").read()
proof_hex = re.search(r'bytes validSignature =\s*hex"([0-9a-fA-F]+)"', sol).group(1)
proof = bytes.fromhex(proof_hex)
assert len(proof) == 2720
open(f"{HERE}/golden_proof_k17_evm.hex", "w").write(proof_hex + "\n")

yul = open(f"{REF}/proving-server/P256Verifier.yul").read()
repr_ = int(re.search(r"mstore\(0x0, (\d+)\)", yul).group(1))
consts = [int(x, 16) for x in re.findall(r"mstore\(0x[0-9a-f]+, 0x([0-9a-f]{64})\)", yul)]
# layout: (1,2) generator, 12 vk commitments as (x, y) pairs, 4 + 4 G2 words
assert consts[0] == 1 and consts[1] == 2 and len(consts) == 2 + 24 + 8
pts = [[consts[2 + 2 * i], consts[3 + 2 * i]] for i in range(12)]

import yul_exec  # noqa: E402

ok, vm = yul_exec.run_verifier(f"{REF}/proving-server/P256Verifier.yul", proof)
assert ok
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
names = ["theta", "beta", "gamma", "y", "x", "v", "u"]
chal = {n: hex(h % R) for n, (_, _, h) in zip(names, vm.keccak_log)}
out = {
    "k": 17,
    "source": "proving-server/P256Verifier.yul (constants), contracts/test/P256Account.t.sol:120-121 (proof)",
    "transcript_repr": hex(repr_),
    "fixed_commitments": [[hex(a), hex(b)] for a, b in pts[:6]],
    "permutation_commitments": [[hex(a), hex(b)] for a, b in pts[6:]],
    "golden_proof_sha256": hashlib.sha256(proof).hexdigest(),
    "golden_challenges": chal,
    "yul_precompile_calls": {str(k): v for k, v in vm.precompile_calls.items()},
}
json.dump(out, open(f"{HERE}/vk_k17.json", "w"), indent=1)
print("ok", chal)
