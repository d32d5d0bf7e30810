#include "wts_io.h"

#include <string.h>

namespace bp {

static void put_matrix(FILE *fp, const char *name, int mrows, int ncols, const float *data)
{
    int stat[5] = {10, mrows, ncols, 0, (int)strlen(name) + 1};
    fwrite(stat, sizeof(int), 5, fp);
    fwrite(name, 1, stat[4], fp);
    fwrite(data, sizeof(float), (size_t)mrows * ncols, fp);
}

void write_weights(FILE *fp, int L, const int *ls, float *const *weights, float *const *bias)
{
    char head[64];
    for (int i = 1; i < L; ++i) {
        snprintf(head, sizeof(head), "weights%d%d", i, i + 1);     // Interface.cc:427
        put_matrix(fp, head, ls[i], ls[i - 1], weights[i]);
        snprintf(head, sizeof(head), "bias%d", i + 1);             // Interface.cc:449
        put_matrix(fp, head, 1, ls[i], bias[i]);
    }
    fflush(fp);
}

std::string read_weights(FILE *fp, int L, const int *ls, float *const *weights, float *const *bias)
{
    int stat[5];
    char head[256];
    for (int i = 1; i < L; ++i) {
        if (fread(stat, sizeof(int), 5, fp) != 5 || stat[4] < 0 || stat[4] > 255 || fread(head, 1, stat[4], fp) != (size_t)stat[4])
            return "init weights file is truncated";
        if (stat[1] != ls[i] || stat[2] != ls[i - 1]) return "init weights node nums do not match";     // Interface.cc:369-374
        const size_t n = (size_t)ls[i - 1] * ls[i];
        if (fread(weights[i], sizeof(float), n, fp) != n) return "init weights file is truncated";
        if (fread(stat, sizeof(int), 5, fp) != 5 || stat[4] < 0 || stat[4] > 255 || fread(head, 1, stat[4], fp) != (size_t)stat[4])
            return "init weights file is truncated";
        if (stat[2] != ls[i] || stat[1] != 1) return "init bias node nums do not match";                 // Interface.cc:379-383
        if (fread(bias[i], sizeof(float), ls[i], fp) != (size_t)ls[i]) return "init weights file is truncated";
    }
    return "";
}

static void rand_weight(float *v, float mn, float mx, size_t n)
{
    for (size_t i = 0; i < n; ++i) v[i] = drand48() * (mx - mn) + mn;
}
void random_weights(int numlayers, const int *layersizes, float *const *weights, float *const *bias, float wmin, float wmax,
                    float bmin, float bmax)
{
    for (int i = 1; i < numlayers; ++i) {
        rand_weight(weights[i], wmin, wmax, (size_t)layersizes[i] * layersizes[i - 1]);
        rand_weight(bias[i], bmin, bmax, (size_t)layersizes[i]);
    }
}
}  // namespace bp
