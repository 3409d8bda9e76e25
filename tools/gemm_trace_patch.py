"""Developer tool: instrument gemm_kernel with s_memtime stamps (prologue / every k-tile / epilogue) -- applies a TEMPORARY
patch to taxoexpan_amd/csrc/txe_gemm.h (undo with `git checkout taxoexpan_amd/csrc/txe_gemm.h`), rebuilds libtxe.so; then
run tools/gemm_trace.py on the GPU box."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(REPO, "taxoexpan_amd", "csrc", "txe_gemm.h")
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b)
rep('''    int row_fast;    // 1: consecutive workgroups walk DOWN a column of tiles (few row panels, many column tiles: the scoring GEMM)
};''', '''    int row_fast;    // 1: consecutive workgroups walk DOWN a column of tiles (few row panels, many column tiles: the scoring GEMM)
    long long* trace;
};''')
rep('''    const int nbn = (N + BN - 1) / BN;
    // XCD-contiguous order over (k-slice, tile)''', '''    long long tr[40]; int trn = 0;
    const bool tracing = T.trace != nullptr && (blockIdx.x % 37 == 5) && blockIdx.y == 0 && blockIdx.x / 37 < 60;
    tr[trn++] = __builtin_readcyclecounter();
    const int nbn = (N + BN - 1) / BN;
    // XCD-contiguous order over (k-slice, tile)''')
rep('''    __syncthreads();
    int t = 1;
    for (; t < nkf; ++t) {                      // tile t (plain) is fetched while tile t-1 is multiplied
        TXE_STAGE_FAST(kbeg + t * GEMM_BK, t & 1, TXE_COMPUTE_TILE((t - 1) & 1))
        __syncthreads();
    }''', '''    __syncthreads();
    tr[trn++] = __builtin_readcyclecounter();
    int t = 1;
    for (; t < nkf; ++t) {                      // tile t (plain) is fetched while tile t-1 is multiplied
        TXE_STAGE_FAST(kbeg + t * GEMM_BK, t & 1, TXE_COMPUTE_TILE((t - 1) & 1))
        __syncthreads();
        if (trn < 30) tr[trn++] = __builtin_readcyclecounter();
    }''')
rep('''    float* Cs = smem;
    __syncthreads();                                 // every wave is done reading the operand stages''', '''    float* Cs = smem;
    tr[trn++] = __builtin_readcyclecounter();
    __syncthreads();                                 // every wave is done reading the operand stages''')
rep('''                    if (n + q < N) epi_store_one(E, m, n + q, v[u][q], cbase);
            }
        }
    }
}

static inline int gcd_vec''', '''                    if (n + q < N) epi_store_one(E, m, n + q, v[u][q], cbase);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    tr[trn++] = __builtin_readcyclecounter();
    if (tracing && threadIdx.x == 0) {
        long long* o = T.trace + (long long)(blockIdx.x / 37) * 48;
        o[0] = trn; o[1] = blockIdx.x;
        for (int i = 0; i < trn; ++i) o[2 + i] = tr[i];
    }
}

static inline int gcd_vec''')
rep('''    T.nfull = 0; T.S = 0; T.ksplit = 0; T.ws = (float*)tail_ws;''', '''    T.nfull = 0; T.S = 0; T.ksplit = 0; T.ws = (float*)tail_ws; T.trace = nullptr;
    { const char* e = getenv("TXE_GEMM_TRACE"); if (e) { T.trace = (long long*)tail_ws; tail_ws = nullptr; T.ws = nullptr; } }''')
open(p, "w").write(s)
subprocess.check_call([sys.executable, os.path.join(REPO, "taxoexpan_amd", "csrc", "build.py")])
