// prove_host — a host that is not Python: the request path of the reference's proving server through the C ABI alone.
//
// What `generate_proof` / `generate_proof_evm` do per request (halo2-circuits/src/ecc/ecdsa_p256.rs:388-428 and :338-378):
// read the ParamsKZG file, read the ProvingKey file, run create_proof over the synthesized advice columns, hand the proof
// bytes back.  Here the same four steps are zk_srs_read, zk_pk_read, zk_poly_upload_canonical + zk_prove, fwrite — what the
// Rust shim of INTEGRATION.md calls, written in plain C++ against include/zkmi355.h (no Python, no torch, no oracle).
// tests/test_gpu_host_example.py drives it and compares its proof with the one the ctypes binding gets.
//
// usage: prove_host <srs.bin> <pk.bin> <advice.bin> <proof.out> k num_advice num_lookup_advice num_fixed lookup_bits idle
//                   <transcript: blake2b|evm> <rng seed: 64 hex digits>
//   srs.bin / pk.bin: SerdeFormat::RawBytes images (zk_srs_write / zk_pk_write, or the Rust host's own files)
//   advice.bin: the advice columns, column-major, 2^k rows of 4 little-endian u64 limbs each (canonical integers)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "zkmi355.h"

static std::vector<uint8_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "prove_host: cannot open %s\n", path);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)len);
    if (len && fread(buf.data(), 1, (size_t)len, f) != (size_t)len) {
        fprintf(stderr, "prove_host: short read of %s\n", path);
        exit(2);
    }
    fclose(f);
    return buf;
}

#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int rc_ = (call);                                                              \
        if (rc_ != ZK_OK) {                                                                  \
            fprintf(stderr, "prove_host: %s failed: %d (hip %d)\n", #call, rc_, ctx ? zk_last_hip_error(ctx) : 0); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 13) {
        fprintf(stderr, "usage: %s srs.bin pk.bin advice.bin proof.out k A L F lookup_bits idle blake2b|evm seedhex\n", argv[0]);
        return 2;
    }
    zk_circuit_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.k = (uint32_t)atoi(argv[5]);
    prm.num_advice = (uint32_t)atoi(argv[6]);
    prm.num_lookup_advice = (uint32_t)atoi(argv[7]);
    prm.num_fixed = (uint32_t)atoi(argv[8]);
    prm.lookup_bits = (uint32_t)atoi(argv[9]);
    prm.num_idle_gate_columns = (uint32_t)atoi(argv[10]);
    const int transcript = strcmp(argv[11], "evm") == 0 ? ZK_TRANSCRIPT_EVM : ZK_TRANSCRIPT_BLAKE2B;
    uint8_t seed[32];
    if (strlen(argv[12]) != 64) {
        fprintf(stderr, "prove_host: the seed is 64 hex digits\n");
        return 2;
    }
    for (int i = 0; i < 32; i++) {
        unsigned v;
        sscanf(argv[12] + 2 * i, "%2x", &v);
        seed[i] = (uint8_t)v;
    }

    zk_ctx* ctx = nullptr;
    CHECK(zk_ctx_create(0, &ctx));  // no device, no proof: there is no CPU path behind this ABI
    {
        const std::vector<uint8_t> srs = slurp(argv[1]);
        CHECK(zk_srs_read(ctx, srs.data(), srs.size(), ZK_SERDE_RAW_BYTES));
    }
    zk_pk pk = 0;
    {
        const std::vector<uint8_t> key = slurp(argv[2]);
        CHECK(zk_pk_read(ctx, &prm, key.data(), key.size(), ZK_SERDE_RAW_BYTES, nullptr, &pk));
    }
    uint32_t shape[8];
    CHECK(zk_pk_shape(ctx, pk, shape));
    const size_t n = (size_t)1 << shape[0], n_adv = shape[2];
    const std::vector<uint8_t> advice = slurp(argv[3]);
    if (advice.size() != n_adv * n * 32) {
        fprintf(stderr, "prove_host: %s holds %zu bytes, the key wants %zu columns of %zu rows\n", argv[3], advice.size(), n_adv, n);
        return 2;
    }
    std::vector<zk_poly> cols(n_adv);
    for (size_t j = 0; j < n_adv; j++) {
        CHECK(zk_poly_alloc(ctx, n, &cols[j]));
        CHECK(zk_poly_upload_canonical(ctx, cols[j], reinterpret_cast<const uint64_t*>(advice.data() + j * n * 32), n));
    }
    size_t len = 0;
    CHECK(zk_proof_size(ctx, pk, transcript, ZK_SCHEME_DEFAULT, &len));
    std::vector<uint8_t> proof(len);
    CHECK(zk_prove(ctx, pk, cols.data(), n_adv, seed, transcript, ZK_SCHEME_DEFAULT, proof.data(), proof.size(), &len));
    FILE* out = fopen(argv[4], "wb");
    if (!out || fwrite(proof.data(), 1, len, out) != len) {
        fprintf(stderr, "prove_host: cannot write %s\n", argv[4]);
        return 2;
    }
    fclose(out);
    for (zk_poly p : cols) zk_poly_free(ctx, p);
    zk_pk_free(ctx, pk);
    zk_ctx_destroy(ctx);
    printf("prove_host: %zu proof bytes\n", len);
    return 0;
}
