// ref_driver.cc -- test infrastructure (build container only): drives the REFERENCE's own `Interface` class
// (compiled from /root/reference/Interface.cc where it lies, see oracle/Makefile target `_ref`) on a synthetic
// Pfile pair and dumps what it hands to the trainer, so that the repo's host code (csrc/host/pfile_reader.cpp,
// wts_io.cpp, bptrain.cpp) can be pinned to reference-generated fixtures (tests/golden/ref_interface_*.npz,
// generator tests/golden/make_ref_fixtures.py).  Nothing of the reference is copied: this file only CALLS its
// public interface (Interface.h:49-62) in the order BPtrain.cc:16-101 does.
//
//   ref_driver epoch <out.bin> name=value ...   one epoch's data path, exactly as BPtrain's main walks it:
//        Initial, get_pfile_info, get_chunk_info(train range), GetRandIndex(chunk order), Readchunk(order[i]) ...,
//        Writeweights (the initial weights as loaded / randomly drawn), get_chunk_info_cv, Readchunk_cv(i) ...
//   out.bin: int32 numlayers, layersizes[numlayers];
//            int32 total_frames, total_sents, framesBeforeSent[total_sents];
//            int32 n_train_chunks, train_total_samples, chunk_frame_st[n], order[n]; per chunk (in read order): int32 n,
//            float in[n*s0], targ[n*sL];
//            int32 n_cv_chunks, cv_total_samples, cv_chunk_frame_st[n]; per chunk: int32 n, float in[], targ[].
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Interface.h"

static void put_i(FILE *o, int v) { fwrite(&v, 4, 1, o); }

int main(int argc, char **argv)
{
    if (argc < 4 || strcmp(argv[1], "epoch") != 0) { printf("usage: ref_driver epoch <out.bin> name=value ...\n"); return 2; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 3;
    Interface *I = new Interface;
    I->Initial(argc - 2, argv + 2);                    // (argv[0] of the shifted vector is skipped by the parser like a program name)
    WorkPara *p = I->para;
    const int L = I->numlayers, s0 = p->layersizes[0], sL = p->layersizes[L - 1];
    put_i(o, L);
    for (int i = 0; i < L; ++i) put_i(o, p->layersizes[i]);
    I->get_pfile_info();
    put_i(o, (int)I->total_frames); put_i(o, (int)I->total_sents);
    fwrite(I->framesBeforeSent, 4, I->total_sents, o);
    I->get_chunk_info(p->train_sent_range);
    const int n = (int)I->total_chunks;
    put_i(o, n); put_i(o, (int)I->total_samples);
    fwrite(I->chunk_frame_st, 4, n, o);
    int *order = new int[n > 0 ? n : 1];
    for (int i = 0; i < n; ++i) order[i] = i;
    I->GetRandIndex(order, n);                         // BPtrain.cc:47
    fwrite(order, 4, n, o);
    for (int i = 0; i < n; ++i) {
        const int cnt = I->Readchunk(order[i]);
        put_i(o, cnt);
        fwrite(p->indata, 4, (size_t)cnt * s0, o);
        fwrite(p->targ, 4, (size_t)cnt * sL, o);
    }
    I->Writeweights();                                 // BPtrain.cc:58 (here: the initial weights, untouched)
    I->get_chunk_info_cv(p->cv_sent_range);
    const int nc = (int)I->cv_total_chunks;
    put_i(o, nc); put_i(o, (int)I->cv_total_samples);
    fwrite(I->cv_chunk_frame_st, 4, nc, o);
    for (int i = 0; i < nc; ++i) {
        const int cnt = I->Readchunk_cv(i);
        put_i(o, cnt);
        fwrite(p->indata, 4, (size_t)cnt * s0, o);
        fwrite(p->targ, 4, (size_t)cnt * sL, o);
    }
    fclose(o);
    return 0;
}
