// prover_batch.h — create_proof for B independent proofs of ONE key in lock-step on one context (zk_prove_batch).
// Included by prover.hip after `struct Prover`; not a translation unit of its own.
//
// Why: a proof alone is a chain of ~150 launches, most of them small (sort heads, reduction tails, scans, staging); several
// independent pipelines overlap those chains, but each still pays them per proof.  The B proofs of a batch — the reference's
// concurrent requests (proving-server/src/main.rs:457-472), the 256 jobs of BASELINE configs[3] — run the same phase at the
// same time, so the same commitment of all of them goes through ONE MSM pass (one sort head, one accumulation launch over
// B x columns, one reduction tail), the same transform through ONE launch per NTT pass (blockIdx.y), all lookups through
// one set of permutation launches, all grand products through one scan, all openings through one evaluation launch.  What
// stays per proof is chip-filling anyway (quotient, linear combinations, Kate divisions) or host work (transcripts, RNG).
//
// Every proof keeps its own transcript, its own ChaCha20 stream and its own workspace (zk_pk_rec::members); the bytes of
// proof j are those of zk_prove with the same key, advice and seed (tests/test_gpu_prove_batch.py) — the phase order below
// is Prover::run's, statement by statement, with the per-proof steps looped and the shared ones merged.

struct BatchRun {
    zk_ctx* c;
    zk_pk_rec* pk0;
    const Layout& lay;
    hipStream_t st;
    uint32_t n, N, B;
    std::vector<Prover*> P;
    BatchBufs& bb;
    uint32_t pass_cap;  // columns per MSM pass
    int rc = ZK_OK;

    BatchRun(zk_ctx* c_, zk_pk_rec* pk_, std::vector<Prover*>& provers, uint32_t cap)
        : c(c_), pk0(pk_), lay(pk_->lay), st(c_->stream), n(pk_->lay.n), N(4 * pk_->lay.n), B((uint32_t)provers.size()), P(provers),
          bb(*pk_->bb), pass_cap(cap) {
        for (Prover* p : P) {
            p->rows = &P[0]->own_rows;  // one row stager for all proofs: one upload + one launch per phase
            p->batch_member = true;
        }
    }

    bool ok() {
        if (rc == ZK_OK)
            for (Prover* p : P)
                if (p->rc != ZK_OK) {
                    rc = p->rc;
                    break;
                }
        return rc == ZK_OK;
    }
    void fail(int code) {
        if (rc == ZK_OK) rc = code;
    }

    // ---- merged commitments: columns of several proofs in one MSM pass; every point goes to its owner's transcript, in the
    // order the columns were given (per proof that is the transcript's order)
    struct Col {
        const Fr* poly;
        uint32_t owner;
    };
    struct Pass {
        int lane;
        std::vector<uint32_t> owners;
    };
    struct Fifo {
        std::vector<int> lanes;
        std::deque<Pass> busy;
        // run before a pass of this queue is collected: what the transcripts must hold first (the lookup passes of a
        // pipelined batch may fill their lanes while the advice pass is still in flight on its own)
        std::function<void()> before_collect;
    };
    void pass_end(const Pass& ps) {
        if (!ok()) return;
        G1Jac js[MSM_MAX_BATCH];
        int r = ctx_msm_end_batch(c, ps.lane, js);
        if (r) return fail(r);
        G1Affine af[MSM_MAX_BATCH];
        const uint32_t cnt = (uint32_t)ps.owners.size();
        jac_batch_to_affine(js, cnt, af);
        for (uint32_t q = 0; q < cnt && ok(); q++)
            if (!P[ps.owners[q]]->tr->write_point(af[q])) fail(ZK_EINVAL);  // identity: halo2 refuses to write it
    }
    void pass_begin(Fifo& f, const std::vector<Col>& cols, int basis) {
        if (!ok() || cols.empty()) return;
        if (f.busy.size() == f.lanes.size()) {
            if (f.before_collect) f.before_collect();
            pass_end(f.busy.front());
            f.busy.pop_front();
        }
        int lane = -1;
        for (int l : f.lanes) {
            bool used = false;
            for (const Pass& ps : f.busy) used = used || ps.lane == l;
            if (!used) lane = l;
        }
        P[0]->rows_flush();
        if (!ok()) return;
        std::vector<const Fr*> polys;
        Pass ps{lane, {}};
        for (const Col& cl : cols) {
            polys.push_back(cl.poly);
            ps.owners.push_back(cl.owner);
        }
        int r = ctx_msm_begin_batch(c, lane, polys.data(), (uint32_t)polys.size(), basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g, n);
        if (r) return fail(r);
        f.busy.push_back(ps);
    }
    void drain(Fifo& f) {
        if (f.before_collect && !f.busy.empty()) f.before_collect();
        while (!f.busy.empty()) {
            pass_end(f.busy.front());
            f.busy.pop_front();
        }
    }
    struct Batcher {
        Fifo* f;
        int basis;
        std::vector<Col> pend;
    };
    void flush(Batcher& b) {
        if (b.pend.size() >= 4 && b.f->lanes.size() >= 2) {
            // two passes on two lanes instead of one: the first half's reduction tail runs under the second half's head
            const size_t h = (b.pend.size() + 1) / 2;
            pass_begin(*b.f, std::vector<Col>(b.pend.begin(), b.pend.begin() + h), b.basis);
            pass_begin(*b.f, std::vector<Col>(b.pend.begin() + h, b.pend.end()), b.basis);
        } else {
            pass_begin(*b.f, b.pend, b.basis);
        }
        b.pend.clear();
    }
    void add(Batcher& b, const Fr* poly, uint32_t owner) {
        b.pend.push_back(Col{poly, owner});
        if (b.pend.size() >= pass_cap) flush(b);
    }
    void transforms(const std::vector<Prover::Forms>& cols) {
        if (!ok()) return;
        P[0]->transforms(cols);  // (flushes the shared row stager first)
    }

    // ------------------------------------------------------------------ run ---
    int run(const Fr* const* advice /* B x n_adv, proof-major */, int scheme) {
        using Forms = Prover::Forms;
        const uint32_t bf = BLINDING_FACTORS, usable = lay.usable, T = 1u << lay.lookup_bits;
        for (Prover* p : P)
            if (p->begin()) return p->rc;
        // the three-coset route (poly.hip): every proof's begin() took the same decision; the members read the key's coset-major
        // copies through their own records
        const bool c3 = P[0]->cosets3;
        if (c3 && (rc = pk_ensure_cosets3(c, pk0))) return rc;

        // -- 1. advice
        const bool many = lay.n_adv > BATCH_ARGS_MIN;
        for (uint32_t q = 0; q < B; q++) {
            zk_pk_rec* pk = P[q]->pk;
            const Fr* const* adv = advice + (size_t)q * lay.n_adv;
            if (many) {  // (every proof stages into its own workspace's argument block)
                CopyPair* h = static_cast<CopyPair*>(pk->h_batch_args);
                for (uint32_t j = 0; j < lay.n_adv; j++) h[j] = CopyPair{adv[j], pk->adv_val[j]};
                if (hipMemcpyAsync(pk->d_batch_args, h, lay.n_adv * sizeof(CopyPair), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
                launch_copy_columns(static_cast<const CopyPair*>(pk->d_batch_args), lay.n_adv, n, st);
            }
            for (uint32_t j = 0; j < lay.n_adv; j++) {
                if (!many) hipMemcpyAsync(pk->adv_val[j], adv[j], (size_t)n * sizeof(Fr), hipMemcpyDeviceToDevice, st);
                P[q]->set_rows(pk->adv_val[j], usable, P[q]->draw(bf + 1));
            }
            P[q]->draw(lay.n_adv);  // advice blinds (unused by KZG, still drawn)
        }
        if (many && aud_sync(c, st) != hipSuccess) return ZK_EHIP;  // the argument staging is reused below
        // one advice column and one lookup per proof (k = 19): the advice pass of all proofs stays in flight on lane 0
        // while the lookup columns are made and committed; otherwise plain order, as Prover::run
        const bool pipe = lay.n_adv == 1 && lay.n_lookups == 1 && B <= pass_cap;
        Fifo af{{0}, {}, nullptr};
        if (pipe) {
            std::vector<Col> cols;
            for (uint32_t q = 0; q < B; q++) cols.push_back(Col{P[q]->pk->adv_val[0], q});
            pass_begin(af, cols, ZK_BASIS_LAGRANGE);
        } else {
            Fifo f{{0, 1, 2}, {}, nullptr};
            std::vector<Col> cols;
            std::vector<Forms> fm;
            auto go = [&]() {
                pass_begin(f, cols, ZK_BASIS_LAGRANGE);
                transforms(fm);
                cols.clear();
                fm.clear();
            };
            for (uint32_t q = 0; q < B && ok(); q++)
                for (uint32_t j = 0; j < lay.n_adv && ok(); j++) {
                    zk_pk_rec* pk = P[q]->pk;
                    cols.push_back(Col{pk->adv_val[j], q});
                    fm.push_back(Forms{pk->adv_val[j], pk->adv_poly[j], pk->adv_coset[j]});
                    if (cols.size() == pass_cap) go();
                }
            if (!cols.empty()) go();
            drain(f);
        }
        if (!ok()) return rc;

        // -- 2. lookups: permuted input / table of every lookup of every proof in one set of launches
        std::vector<Fr> theta(B, Fr::zero());
        bool theta_done = false;
        auto squeeze_theta = [&]() {
            if (!theta_done) {
                if (pipe) drain(af);
                for (uint32_t q = 0; q < B; q++) theta[q] = P[q]->tr->squeeze();
                theta_done = true;
            }
        };
        hipMemsetAsync(bb.lks.err, 0, 4, st);
        // (no a' / s' commitment reaches a transcript before the proof's advice commitments and theta)
        Fifo lf{pipe ? std::vector<int>{1, 2} : std::vector<int>{0, 1, 2}, {}, squeeze_theta};
        Batcher lb{&lf, ZK_BASIS_LAGRANGE, {}};
        std::vector<Forms> due;
        if (!pipe) squeeze_theta();  // the advice commitments are all written: theta precedes the first a'
        {
            const uint32_t total = B * lay.n_lookups;
            for (uint32_t i0 = 0; i0 < total; i0 += MAX_LOOKUPS) {  // (LkPtrs holds MAX_LOOKUPS lookups)
                LkPtrs lp;
                memset(&lp, 0, sizeof(lp));
                const uint32_t cnt = std::min<uint32_t>(MAX_LOOKUPS, total - i0);
                for (uint32_t i = 0; i < cnt; i++) {
                    const uint32_t q = (i0 + i) / lay.n_lookups, l = (i0 + i) % lay.n_lookups;
                    zk_pk_rec* pk = P[q]->pk;
                    if (lay.single) {
                        launch_mul(pk->lk_in[l], pk->fixed_val[lay.fx_qlookup], pk->adv_val[0], n, st);
                        lp.inp[i] = pk->lk_in[l];
                    } else {
                        lp.inp[i] = pk->adv_val[lay.n_gate + l];
                    }
                    lp.ap[i] = pk->lk_ap[l];
                    lp.sp[i] = pk->lk_sp[l];
                }
                LookupScratch s = bb.lks;  // this group's slice of the scratch: lookup i0's arrays first
                s.hist += (size_t)i0 * s.stride;
                s.present += (size_t)i0 * s.stride;
                s.absent += (size_t)i0 * s.stride;
                s.off += (size_t)i0 * s.stride;
                s.dex += (size_t)i0 * s.stride;
                s.aex += (size_t)i0 * s.stride;
                s.bsum += (size_t)i0 * s.stride;
                launch_lookup_permute(lp, cnt, usable, T, s, st);
            }
        }
        for (uint32_t q = 0; q < B && ok(); q++) {
            zk_pk_rec* pk = P[q]->pk;
            for (uint32_t l = 0; l < lay.n_lookups && ok(); l++) {
                P[q]->set_rows(pk->lk_ap[l], usable, P[q]->draw(bf + 1));
                P[q]->set_rows(pk->lk_sp[l], usable, P[q]->draw(bf + 1));
                P[q]->draw(2);
                add(lb, pk->lk_ap[l], q);
                add(lb, pk->lk_sp[l], q);
                due.push_back(Forms{pk->lk_ap[l], pk->lk_ap_poly[l], pk->lk_ap_coset[l]});
                due.push_back(Forms{pk->lk_sp[l], pk->lk_sp_poly[l], pk->lk_sp_coset[l]});
                if (lb.pend.empty()) {
                    transforms(due);
                    due.clear();
                }
            }
        }
        flush(lb);
        if (pipe)
            for (uint32_t q = 0; q < B; q++) due.push_back(Forms{P[q]->pk->adv_val[0], P[q]->pk->adv_poly[0], P[q]->pk->adv_coset[0]});
        transforms(due);
        due.clear();
        {
            // one check for all lookups of all proofs (the flag accumulates): an input outside the table is halo2's
            // ConstraintSystemFailure; nothing has been written for the lookups yet.  The batch fails as a whole.
            uint32_t* err = reinterpret_cast<uint32_t*>(c->host_small);
            if (hipMemcpyAsync(err, bb.lks.err, 4, hipMemcpyDeviceToHost, st) != hipSuccess || aud_sync(c, st) != hipSuccess)
                return ZK_EHIP;
            if (*err) {
                ctx_msm_drain(c);
                return ZK_EWITNESS;
            }
        }
        squeeze_theta();
        drain(lf);
        if (!ok()) return rc;
        std::vector<Fr> beta(B), gamma(B);
        for (uint32_t q = 0; q < B; q++) {
            beta[q] = P[q]->tr->squeeze();
            gamma[q] = P[q]->tr->squeeze();
        }

        // -- 5 (early). the random polynomials: no challenge needed; their n draws come after the grand products' draws
        Fifo rf{{0}, {}, nullptr};
        {
            const uint64_t skip = (uint64_t)lay.n_chunks * (bf + 1) + (uint64_t)lay.n_lookups * (bf + 1);
            std::vector<Col> cols;
            for (uint32_t q = 0; q < B; q++) {
                ChaChaKey key;
                memcpy(key.w, P[q]->rng.key, 32);
                launch_chacha_fr(key, P[q]->rng.block + skip, P[q]->pk->random_poly, n, st);
                cols.push_back(Col{P[q]->pk->random_poly, q});
            }
            // (more proofs than a pass takes: the surplus random polynomials are committed after the grand products)
            if (B <= pass_cap) pass_begin(rf, cols, ZK_BASIS_MONOMIAL);
        }

        // -- 3. grand products of every proof: numerators / denominators per proof, then ALL scans in one batch
        Fifo zf{{1, 2}, {}, nullptr};
        Batcher zb{&zf, ZK_BASIS_LAGRANGE, {}};
        std::vector<Forms> zdue;
        const uint32_t nprod = lay.n_chunks + lay.n_lookups, np = B * nprod;
        {
            std::vector<GpItem> items(np);
            std::vector<Fr*> zs(np);
            const uint32_t nblk = gp_blocks(n);
            const Fr delta = fr_delta();
            const bool many_chunks = lay.n_chunks > BATCH_ARGS_MIN, many_lookups = lay.n_lookups > BATCH_ARGS_MIN;
            if ((many_chunks || many_lookups) && aud_sync(c, st) != hipSuccess) return ZK_EHIP;  // the argument staging may still be in use
            for (uint32_t q = 0; q < B; q++) {
                zk_pk_rec* pk = P[q]->pk;
                Fr dcur = Fr::one();
                for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
                    PermArgs a;
                    memset(&a, 0, sizeof(a));
                    a.n = n;
                    const uint32_t lo = ci * lay.chunk_len, hi = std::min<uint32_t>((uint32_t)lay.perm_cols.size(), lo + lay.chunk_len);
                    a.ncols = hi - lo;
                    for (uint32_t p = lo; p < hi; p++) {
                        a.values[p - lo] = P[q]->col_val(lay.perm_cols[p]);
                        a.sigma[p - lo] = pk->sigma_val[p];
                        a.delta[p - lo] = dcur;
                        dcur = fe_mul(dcur, delta);
                    }
                    a.tw = P[q]->tw;
                    a.beta = beta[q];
                    a.gamma = gamma[q];
                    a.num = pk->gp_num[ci];
                    a.den = pk->gp_den[ci];
                    if (many_chunks) static_cast<PermArgs*>(pk->h_batch_args)[ci] = a;
                    else launch_perm_numden(a, st);
                    zs[q * nprod + ci] = pk->z_val[ci];
                }
                if (many_chunks) {
                    if (hipMemcpyAsync(pk->d_batch_args, pk->h_batch_args, lay.n_chunks * sizeof(PermArgs), hipMemcpyHostToDevice, st) != hipSuccess)
                        return ZK_EHIP;
                    launch_perm_numden_batch(static_cast<const PermArgs*>(pk->d_batch_args), lay.n_chunks, n, st);
                    if (many_lookups && aud_sync(c, st) != hipSuccess) return ZK_EHIP;  // the staging is rewritten below
                }
                for (uint32_t l = 0; l < lay.n_lookups; l++) {
                    const Fr* inp = lay.single ? pk->lk_in[l] : pk->adv_val[lay.n_gate + l];
                    const uint32_t p = lay.n_chunks + l;
                    if (many_lookups)
                        static_cast<LkNumDenArgs*>(pk->h_batch_args)[l] =
                            LkNumDenArgs{pk->lk_ap[l], pk->lk_sp[l], inp, pk->fixed_val[lay.fx_table], pk->gp_num[p], pk->gp_den[p]};
                    else
                        launch_lk_numden(pk->lk_ap[l], pk->lk_sp[l], inp, pk->fixed_val[lay.fx_table], beta[q], gamma[q], pk->gp_num[p],
                                         pk->gp_den[p], n, st);
                    zs[q * nprod + p] = pk->lk_z[l];
                }
                if (many_lookups) {
                    if (hipMemcpyAsync(pk->d_batch_args, pk->h_batch_args, lay.n_lookups * sizeof(LkNumDenArgs), hipMemcpyHostToDevice, st) !=
                        hipSuccess)
                        return ZK_EHIP;
                    launch_lk_numden_batch(static_cast<const LkNumDenArgs*>(pk->d_batch_args), lay.n_lookups, beta[q], gamma[q], n, st);
                }
                for (uint32_t p = 0; p < nprod; p++) {
                    GpItem& it = items[q * nprod + p];
                    it.num = pk->gp_num[p];
                    it.den = pk->gp_den[p];
                    it.loc_p = pk->gp_loc_p[p];
                    it.loc_r = pk->gp_loc_r[p];
                    it.tot_p = pk->gp_tot + (size_t)2 * nblk * p;
                    it.tot_r = it.tot_p + nblk;
                    it.z = zs[q * nprod + p];
                    it.chain = (p > 0 && p < lay.n_chunks) ? 1u : 0u;  // chunk ci starts from chunk ci-1's z at row `usable`; a proof's first product starts a new chain
                    it.pad_ = 0;
                }
            }
            Fr* q_dev = bb.gp_scal;
            Fr* qinv_dev = bb.gp_scal + np;
            Fr* k_dev = bb.gp_scal + 2 * (size_t)np;
            Fr* init_dev = bb.gp_scal + 3 * (size_t)np;
            bool fast = !c->opt_gp_batch_invert;  // zk_ctx_set_option(ZK_OPT_GP_BATCH_INVERT)
            if (fast) {
                if (hipMemcpyAsync(bb.d_gp_items, items.data(), np * sizeof(GpItem), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
                launch_gp_batch_scan(bb.d_gp_items, np, n, q_dev, st);
                if (hipMemcpyAsync(bb.gp_host, q_dev, np * sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    aud_sync(c, st) != hipSuccess)
                    return ZK_EHIP;
                // all inverses of all proofs with one field inversion
                Fr* qv = bb.gp_host;
                Fr* qi = bb.gp_host + np;
                Fr run = Fr::one();
                for (uint32_t p = 0; p < np && fast; p++) {
                    if (qv[p].is_zero()) fast = false;
                    qi[p] = run;
                    run = fe_mul(run, qv[p]);
                }
                if (fast) {
                    Fr inv = fe_inv_fast(run);
                    for (uint32_t p = np; p-- > 0;) {
                        const Fr t = fe_mul(inv, qi[p]);
                        inv = fe_mul(inv, qv[p]);
                        qi[p] = t;
                    }
                    if (hipMemcpyAsync(qinv_dev, qi, np * sizeof(Fr), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
                    launch_gp_batch_apply(bb.d_gp_items, np, n, usable, qinv_dev, k_dev, init_dev, st);
                }
            }
            if (!fast) {
                // a zero denominator somewhere: halo2's batch_invert semantics (0 -> 0), product by product, every proof
                for (uint32_t q = 0; q < B; q++) {
                    zk_pk_rec* pk = P[q]->pk;
                    for (uint32_t p = 0; p < nprod; p++) {
                        launch_frac(pk->gp_num[p], pk->gp_den[p], pk->t_frac, n, st);
                        const Fr* prev = items[q * nprod + p].chain ? zs[q * nprod + p - 1] + usable : nullptr;
                        launch_prefix_product(pk->t_frac, zs[q * nprod + p], n, prev, Fr::one(), pk->t_a, pk->t_small, st);
                    }
                }
            }
            // blinding rows and commitments, per proof in halo2's order (chunks, then lookups)
            for (uint32_t q = 0; q < B && ok(); q++) {
                zk_pk_rec* pk = P[q]->pk;
                for (uint32_t p = 0; p < nprod && ok(); p++) {
                    Fr* z = zs[q * nprod + p];
                    P[q]->set_rows(z, n - bf, P[q]->draw(bf));
                    P[q]->draw(1);
                    add(zb, z, q);
                    if (p < lay.n_chunks) zdue.push_back(Forms{pk->z_val[p], pk->z_poly[p], pk->z_coset[p]});
                    else zdue.push_back(Forms{pk->lk_z[p - lay.n_chunks], pk->lk_z_poly[p - lay.n_chunks], pk->lk_z_coset[p - lay.n_chunks]});
                    if (zb.pend.empty()) {
                        transforms(zdue);
                        zdue.clear();
                    }
                }
            }
        }
        flush(zb);
        transforms(zdue);
        zdue.clear();
        drain(zf);
        if (!ok()) return rc;

        // -- 5. the random polynomials' commitments (their draws happen here in stream order)
        for (uint32_t q = 0; q < B; q++) {
            P[q]->rng.block += n;
            P[q]->draw(1);
        }
        if (B <= pass_cap) {
            drain(rf);
        } else {
            Fifo f{{0, 1, 2}, {}, nullptr};
            Batcher rb{&f, ZK_BASIS_MONOMIAL, {}};
            for (uint32_t q = 0; q < B; q++) add(rb, P[q]->pk->random_poly, q);
            flush(rb);
            drain(f);
        }
        if (!ok()) return rc;
        std::vector<Fr> y(B);
        for (uint32_t q = 0; q < B; q++) y[q] = P[q]->tr->squeeze();

        // -- 6. quotients: one launch per proof (each fills the chip), then the inverse coset transforms of all in batches
        for (uint32_t q = 0; q < B; q++) {
            Prover& pr = *P[q];
            QuotientCosets qc;
            zk_pk_rec* pk = pr.pk;
            for (uint32_t j = 0; j < lay.n_adv; j++) qc.adv.push_back(pk->adv_coset[j]);
            for (uint32_t ci = 0; ci < lay.n_chunks; ci++) qc.z.push_back(pk->z_coset[ci]);
            for (uint32_t l = 0; l < lay.n_lookups; l++) {
                qc.lk_a.push_back(pk->lk_ap_coset[l]);
                qc.lk_s.push_back(pk->lk_sp_coset[l]);
                qc.lk_z.push_back(pk->lk_z_coset[l]);
            }
            qc.cosets3 = c3;
            if (int r = pk_quotient(c, pk, qc, beta[q], gamma[q], y[q], true, pk->h_ext)) return r;
        }
        if (c3) {
            for (uint32_t q = 0; q < B; q++)
                if (int r = ctx_intt_cosets3(c, P[q]->pk->h_ext, lay.k)) return r;
        } else {
            const uint32_t b2 = ctx_ntt_max_batch(lay.ext_k);
            const Fr* src[NTT_MAX_BATCH];
            Fr* dst[NTT_MAX_BATCH];
            for (uint32_t q0 = 0; q0 < B; q0 += b2) {
                const uint32_t cnt = std::min(b2, B - q0);
                for (uint32_t i = 0; i < cnt; i++) src[i] = dst[i] = P[q0 + i]->pk->h_ext;
                if (int r = ctx_ntt_batch(c, src, N, dst, cnt, lay.ext_k, true, true, N)) return r;
            }
        }
        {
            // the h pieces of every proof: contiguous n-coefficient slices of its quotient
            Fifo hf{{0, 1, 2}, {}, nullptr};
            Batcher hb{&hf, ZK_BASIS_MONOMIAL, {}};
            for (uint32_t q = 0; q < B && ok(); q++) {
                P[q]->draw(lay.n_h);  // h-piece blinds
                for (uint32_t i = 0; i < lay.n_h && ok(); i++) add(hb, P[q]->pk->h_ext + (size_t)i * n, q);
            }
            flush(hb);
            drain(hf);
        }
        if (!ok()) return rc;
        std::vector<Fr> x(B);
        for (uint32_t q = 0; q < B; q++) x[q] = P[q]->tr->squeeze();

        // -- 7. evaluations: every opened value of every proof in ONE launch
        std::vector<std::vector<Prover::Q>> ev(B);
        std::vector<Prover::EvIdx> ix(B);
        std::vector<std::vector<Prover::Q>> queries(B);
        {
            size_t total = 0;
            for (uint32_t q = 0; q < B; q++) {
                P[q]->combine_h(x[q]);
                P[q]->build_evals(ev[q], ix[q]);
                if (ev[q].size() > pk0->max_evals) return ZK_ESTATE;
                for (size_t i = 0; i < ev[q].size(); i++) {
                    bb.h_evargs[total + i].poly = ev[q][i].poly;
                    bb.h_evargs[total + i].x = P[q]->xrot(x[q], ev[q][i].rot);
                }
                total += ev[q].size();
            }
            hipEventRecord(c->ev[ZK_T_EVAL][0], st);
            launch_eval_batch(bb.h_evargs, bb.d_evargs, (uint32_t)total, n, bb.ev_scratch, bb.ev_out, st);
            hipEventRecord(c->ev[ZK_T_EVAL][1], st);
            c->ev_valid[ZK_T_EVAL] = true;
            if (hipMemcpyAsync(bb.tail_host, bb.ev_out, total * sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess ||
                aud_sync(c, st) != hipSuccess)
                return ZK_EHIP;
            size_t pos = 0;
            for (uint32_t q = 0; q < B; q++) {
                for (size_t i = 0; i < ev[q].size(); i++) ev[q][i].eval = bb.tail_host[pos + i];
                pos += ev[q].size();
                for (size_t i = 0; i < ix[q].n_written; i++) P[q]->tr->write_scalar(ev[q][i].eval);
                queries[q] = P[q]->queries_from_evals(ev[q], ix[q]);
                P[q]->pk->lc_used = 0;
            }
        }
        if (!ok()) return rc;

        // -- 8. multi-open: the stages of every proof, the commitments between them merged
        if (scheme == ZK_SCHEME_GWC) {
            Fifo wf{{0, 1, 2}, {}, nullptr};
            Batcher wb{&wf, ZK_BASIS_MONOMIAL, {}};
            for (uint32_t q = 0; q < B && ok(); q++) {
                std::vector<const Fr*> wit;
                if (int r = P[q]->gwc_stage1(queries[q], x[q], wit)) return r;
                for (const Fr* w : wit) add(wb, w, q);
            }
            flush(wb);
            drain(wf);
            return ok() ? ZK_OK : rc;
        }
        std::vector<Prover::Shplonk> S(B);
        Fifo of{{0, 1, 2}, {}, nullptr};
        {
            Batcher ob{&of, ZK_BASIS_MONOMIAL, {}};
            for (uint32_t q = 0; q < B && ok(); q++) {
                if (int r = P[q]->shplonk_stage1(queries[q], x[q], S[q])) return r;
                add(ob, S[q].hx, q);
            }
            flush(ob);
            drain(of);
        }
        if (!ok()) return rc;
        {
            Batcher ob{&of, ZK_BASIS_MONOMIAL, {}};
            for (uint32_t q = 0; q < B && ok(); q++) {
                const Fr* last = nullptr;
                if (int r = P[q]->shplonk_stage2(S[q], &last)) return r;
                add(ob, last, q);
            }
            flush(ob);
            drain(of);
        }
        return ok() ? ZK_OK : rc;
    }
};
