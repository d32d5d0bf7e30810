// tests/cpp/shim_caller.cc -- a BPtrain.cc-shaped caller (BPtrain.cc:31-32,53,57,77,96) compiled
// against include/BP_GPU.h with a plain C++ compiler and linked with libbp_hip.so.
// usage: shim_caller <in.bin> <out.bin>
//   in.bin : int32 numlayers, layersizes[numlayers], bunchsize, dropoutflag, n_train, n_cv;
//            float32 lrate, momentum, weightcost, visible_omit, hid_omit;
//            then weights[1..L-1], bias[1..L-1], train_in, train_targ, cv_in, cv_targ (float32)
//   out.bin: float32 cv_squared_error, then weights[1..L-1], bias[1..L-1] after training
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "BP_GPU.h"

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { printf("short read\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int L; rd(f, &L, 4);
    int ls[MAXLAYER] = {0}; rd(f, ls, 4 * L);
    int bunch, dropoutflag, n_train, n_cv; rd(f, &bunch, 4); rd(f, &dropoutflag, 4); rd(f, &n_train, 4); rd(f, &n_cv, 4);
    float hp[5]; rd(f, hp, 20);
    float *weights[MAXLAYER] = {0}, *bias[MAXLAYER] = {0};
    for (int l = 1; l < L; ++l) { weights[l] = new float[(size_t)ls[l - 1] * ls[l]]; rd(f, weights[l], 4 * (size_t)ls[l - 1] * ls[l]); }
    for (int l = 1; l < L; ++l) { bias[l] = new float[ls[l]]; rd(f, bias[l], 4 * (size_t)ls[l]); }
    std::vector<float> tin((size_t)n_train * ls[0]), ttg((size_t)n_train * ls[L - 1]), cin((size_t)n_cv * ls[0]), ctg((size_t)n_cv * ls[L - 1]);
    rd(f, tin.data(), 4 * tin.size()); rd(f, ttg.data(), 4 * ttg.size());
    rd(f, cin.data(), 4 * cin.size()); rd(f, ctg.data(), 4 * ctg.size());
    fclose(f);

    BP_GPU *TrainObj = new BP_GPU(1, L, ls, bunch, hp[0], hp[1], hp[2], weights, bias, dropoutflag, hp[3], hp[4]);
    TrainObj->train(n_train, tin.data(), ttg.data());
    TrainObj->returnWeights(weights, bias);
    float squared_err = TrainObj->CrossValid(n_cv, cin.data(), ctg.data());
    delete TrainObj;

    FILE *o = fopen(argv[2], "wb");
    fwrite(&squared_err, 4, 1, o);
    for (int l = 1; l < L; ++l) fwrite(weights[l], 4, (size_t)ls[l - 1] * ls[l], o);
    for (int l = 1; l < L; ++l) fwrite(bias[l], 4, ls[l], o);
    fclose(o);
    printf("all finish!\n");
    return 1;   // BPtrain.cc:100 returns 1 on success
}
