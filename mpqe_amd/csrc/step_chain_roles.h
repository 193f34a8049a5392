// The body of the chain launch's kernels (step.hip: step_chain_kernel, step_chain_fwd_kernel) -- a FRAGMENT, included inside
// each of them, not a header of declarations. In scope: the kernel's arguments sd, lp, tabs, ca, pa, po, its __shared__
// ChainLds<NCB, KS, NW> S and its template arguments; CHAIN_ROLES_RETURN is `return` in the whole step's kernel and a jump to
// its finish in the forward-only step's. (As an inlined function taking the arguments by reference the same code ran
// 2.3 us slower in the whole step's kernel -- 42.8 against 40.4 us, same box, same registers: measured in round 5.)
//
// role of this workgroup (uniform): chain workgroup, prologue work in front of / behind them, zero fill, then the
// post roles: a producer is never queued behind a consumer that waits for it
    int bid = (int)blockIdx.x, role;
    if (bid == 0 && threadIdx.x == 0 && pa.tail_arrive) *pa.tail_arrive = 0u;       // (read by the NEXT launch)
    if (bid == 0 && threadIdx.x == 0 && pa.runs_count) *pa.runs_count = 0;
    if (pa.strail && bid >= (int)gridDim.x - pa.strail) {
        if (NW == 4 || threadIdx.x < TSORT_THREADS) tsort_block(pa.ts, bid - ((int)gridDim.x - pa.strail), reinterpret_cast<unsigned *>(S.xs));
        CHAIN_ROLES_RETURN;
    }
    if (pa.plast) {
        if (bid < pa.sblocks) role = 1;
        else if (bid < pa.sblocks + pa.nchain) role = 0, bid -= pa.sblocks;
        else if (bid < pa.lead + pa.nchain) {
            const int t = bid - pa.sblocks - pa.nchain;
            const int rk = (int)((pa.plxrank >> (4 * (t & 7))) & 15u) - 1;
            if (rk < 0) CHAIN_ROLES_RETURN;
            role = 1, bid = pa.sblocks + (t >> 3) * pa.plna + rk;      // (numbered on from the items the sort rows held)
        } else role = 2, bid -= pa.lead + pa.nchain;
    } else if (bid < pa.lead) role = 1;
    else if (bid < pa.lead + pa.nchain) role = 0, bid -= pa.lead;
    else role = 2, bid -= pa.lead + pa.nchain;
    if (role == 0) {
#ifndef MPQE_EMU
        if (po.zpad > 0) __builtin_amdgcn_s_setprio(1);      // (merged launch: over the tiles that may share the CU)
#endif
        ca.cb = bid;
        ca.nchain = pa.nchain;
        chain_block<NCB, KS, NW, RO>(sd, lp, tabs, ca, S);
    } else if (role == 1) {
        constexpr int D = 16 * NCB * NW / KS;
        // The touch plan of THIS step's ids (step_touch.h): the first workgroups of the launch, so all of them are
        // resident before any other workgroup is dispatched (they synchronise among themselves); nothing in the launch
        // waits for them -- the plan is read by the step's last launch.
        if (bid < pa.sblocks) {
            static_assert(sizeof(S.xs) >= TSORT_LDS_WORDS * sizeof(unsigned), "the sort's LDS tables live in the state buffers");
            const int rk = (int)((pa.sxrank >> (4 * (bid & 7))) & 15u) - 1;
            if (rk >= 0) {
                const int sb = (bid >> 3) * pa.sna + rk;
                if (sb < pa.ts.nblk && (NW == 4 || threadIdx.x < TSORT_THREADS))
                    tsort_block(pa.ts, sb, reinterpret_cast<unsigned *>(S.xs));
                CHAIN_ROLES_RETURN;
            }
            // (not one of the sort's XCDs: the next prologue item -- no hole in front of the chain workgroups of the XCDs
            // whose CUs are all needed; its rank among the other XCDs)
            int orank = 0;
            for (int x = 0; x < (bid & 7); ++x) orank += ((pa.sxrank >> (4 * x)) & 15u) == 0u;
            bid = (bid >> 3) * (8 - pa.sna) + orank;
        } else {
            bid -= pa.sblocks / 8 * pa.sna;               // (items the rows above have taken: sblocks / 8 x (8 - sna))
        }
        if (bid < pa.ublocks) {
            if (NW == 4 || threadIdx.x < 256) uop_block(bid, D, lp, pa.ua, S.xs, nullptr, 0);
            // merged launch: the post-pass reads the pre-pass' vectors from VT. Wave 0 made the stores (write-through):
            // once they are acknowledged the workgroup counts itself in (uop_wait_prepass)
            if (pa.fwd_done && threadIdx.x < 64) {
#ifndef MPQE_EMU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                if (threadIdx.x == 0) atomicAdd(pa.fwd_done, 1u);
            }
        } else if (bid < pa.ublocks + pa.tblocks) {
#ifndef MPQE_EMU
            if (pa.late && bid == pa.ublocks)
                for (int q = 0; q < (1 << 18); ++q) __builtin_amdgcn_s_sleep(127);      // (~1 s: 2^18 x 8 128 cycles; uniform)
#endif
            prep_transpose_block(lp, pa, D, bid - pa.ublocks, S.xs);
        }       // (else: padding)
    } else if (po.zpad == 0 || bid < po.zpad) {
        if ((long long)bid < pa.zs.block0[pa.zs.count]) prep_zero_block(pa.zs, bid);
        // (split tail: the relation matrices nothing writes this step are zero-filled HERE, behind the chain workgroups -- the
        // launch has idle CUs from the moment its light batches are through -- not by workgroups of the weight-gradient launch)
        else if (po.zpad == 0 && po.zmblocks > 0 && (long long)bid < pa.zs.block0[pa.zs.count] + po.zmblocks) {
            if (NW == 4 || threadIdx.x < 256) zmat_block(po.zmats, po.zper, bid - (int)pa.zs.block0[pa.zs.count], po.D, po.gp);
        }
    } else {
        // (post roles only on the XCDs picked for them: workgroup b runs on XCD b % 8; the others leave at once)
        const int pb = bid - po.zpad;
        const int rk = (int)((po.xrank >> (4 * (pb & 7))) & 15u) - 1;
        if (rk >= 0 && (NW == 4 || threadIdx.x < 256))
            post_block<(sizeof(S) >= 4 * 64 * GWR_LDT * sizeof(float)) ? 4 : 1>(sd, lp, po, (pb >> 3) * po.na + rk,
                                                                                reinterpret_cast<float *>(&S));
    }
