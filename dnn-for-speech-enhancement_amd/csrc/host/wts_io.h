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
}  // namespace bp
