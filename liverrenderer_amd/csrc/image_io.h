// Image readers the liver scenes need: 8/16-bit PNG (bump map) and scanline
// OpenEXR with NONE / ZIPS / ZIP / PIZ compression (environment map).
// Replaces libpng / OpenEXR as used by src/core/bitmap.cpp in the reference
// (both absent from the build image).  Also a minimal uncompressed EXR writer.
#pragma once
#include <string>
#include <vector>
#include <cstdint>

namespace lrt {

struct Image {
    int width = 0, height = 0, channels = 0;
    int bits_per_channel = 0;          // of the file (8, 16, 32)
    bool srgb = false;                 // PNG: values are gamma-encoded [0,1]
    std::vector<std::string> channel_names;   // EXR
    std::vector<float> data;           // height * width * channels
};

Image read_png(const std::string &path);                  // throws std::runtime_error
Image read_exr(const std::string &path);                  // channels in file (alphabetical) order
Image read_image_rgb(const std::string &path);            // by extension; RGB(A)/Y -> channels as stored
void  write_exr(const std::string &path, int w, int h, int channels, const float *data);  // RGB / RGBA float32, no compression
void  write_png(const std::string &path, int w, int h, int channels, const float *data);  // 8-bit, sRGB-encoded colour, linear alpha

} // namespace lrt
