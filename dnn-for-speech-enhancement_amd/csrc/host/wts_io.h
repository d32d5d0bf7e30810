// wts_io.h -- the reference's weight file (a sequence of MATLAB Level-4 MAT matrices, little endian,
// type 10 = single precision full matrix): Interface.cc:362-387 (read), :411-465 (write).
#pragma once
#include <stdio.h>
#include <string>

namespace bp {
// weights[l] ([prev][cur] row-major == column-major cur x prev), bias[l], l = 1..numlayers-1
void write_weights(FILE *fp, int numlayers, const int *layersizes, float *const *weights, float *const *bias);
// returns an empty string on success, else the reference's log message
std::string read_weights(FILE *fp, int numlayers, const int *layersizes, float *const *weights, float *const *bias);
// random initial net when no initwts_file is given (Interface.cc:340-349 + GetRandWeight :1036-1042): per layer the
// weights then the biases, each value drand48() * (max - min) + min, consuming the process-wide srand48 stream
void random_weights(int numlayers, const int *layersizes, float *const *weights, float *const *bias, float wmin, float wmax,
                    float bmin, float bmax);
}  // namespace bp
