// dvo_benchmark/image_io.h -- what the reference's benchmark drivers get from cv::imread + cv::cvtColor +
// SurfacePyramid::convertRawDepthImageSse when they load one TUM frame (dvo_benchmark/src/benchmark_slam.cpp:46-93,
// dvo_core/src/core/surface_pyramid.cpp:65-105), without OpenCV: a PNG decoder on zlib (non-interlaced, 8/16-bit,
// grey / RGB / RGBA), OpenCV's fixed-point BGR->grey, and the raw-depth conversion.  Link with -lz.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "dvo/compat.h"

namespace dvo_benchmark {

struct PngImage {
  int width, height, channels, bit_depth;
  std::vector<uint8_t> bytes;   // rows top-down, samples interleaved, 16-bit samples big-endian as stored in the file
  PngImage() : width(0), height(0), channels(0), bit_depth(0) {}
  bool empty() const { return width == 0 || height == 0; }
  unsigned sample(int x, int y, int c) const {
    const size_t i = (size_t(y) * width + x) * channels + c;
    return bit_depth == 16 ? (unsigned(bytes[2 * i]) << 8) | bytes[2 * i + 1] : bytes[i];
  }
};

namespace detail {

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace detail

// An empty image when the file cannot be opened (cv::imread's contract); throws on a file that is not a PNG this
// decoder handles.
inline PngImage readPng(const std::string& path) {
  PngImage img;
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return img;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
  std::fclose(f);
  static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (file.size() < 8 || std::memcmp(file.data(), magic, 8) != 0) throw std::runtime_error(path + ": not a PNG file");

  std::vector<uint8_t> idat;
  int colour_type = -1;
  for (size_t pos = 8; pos + 12 <= file.size();) {
    const uint32_t len = detail::be32(&file[pos]);
    const uint8_t* tag = &file[pos + 4];
    const uint8_t* body = &file[pos + 8];
    if (pos + 12 + len > file.size()) throw std::runtime_error(path + ": truncated PNG chunk");
    if (std::memcmp(tag, "IHDR", 4) == 0 && len >= 13) {
      img.width = int(detail::be32(body));
      img.height = int(detail::be32(body + 4));
      img.bit_depth = body[8];
      colour_type = body[9];
      if (body[12] != 0) throw std::runtime_error(path + ": interlaced PNG not supported");
    } else if (std::memcmp(tag, "IDAT", 4) == 0) {
      idat.insert(idat.end(), body, body + len);
    } else if (std::memcmp(tag, "IEND", 4) == 0) {
      break;
    }
    pos += 12 + size_t(len);
  }
  switch (colour_type) {
    case 0: img.channels = 1; break;
    case 2: img.channels = 3; break;
    case 4: img.channels = 2; break;
    case 6: img.channels = 4; break;
    default: throw std::runtime_error(path + ": unsupported PNG colour type");
  }
  if (img.bit_depth != 8 && img.bit_depth != 16) throw std::runtime_error(path + ": unsupported PNG bit depth");

  const size_t bpp = size_t(img.channels) * img.bit_depth / 8, stride = bpp * img.width;
  std::vector<uint8_t> raw((stride + 1) * img.height);
  uLongf raw_len = uLongf(raw.size());
  if (uncompress(raw.data(), &raw_len, idat.data(), uLong(idat.size())) != Z_OK || raw_len != raw.size())
    throw std::runtime_error(path + ": PNG data does not inflate to the declared size");

  img.bytes.assign(stride * img.height, 0);
  for (int y = 0; y < img.height; ++y) {
    const uint8_t ftype = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* out = &img.bytes[stride * y];
    const uint8_t* up = y > 0 ? out - stride : 0;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred;
      switch (ftype) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: pred = detail::paeth(a, b, c); break;
        default: throw std::runtime_error(path + ": bad PNG filter type");
      }
      out[i] = uint8_t(in[i] + pred);
    }
  }
  return img;
}

// cv::cvtColor(CV_BGR2GRAY) on 8-bit data: fixed-point ITU-R BT.601 with 14 fractional bits
inline uint8_t greyFromRgb8(unsigned r, unsigned g, unsigned b) { return uint8_t((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14); }

// the grey float image (0..255) `load` hands to RgbdCameraPyramid::create (benchmark_slam.cpp:55-69)
inline dvo::compat::ImageMat greyFloatFromPng(const PngImage& png) {
  if (png.bit_depth != 8) throw std::runtime_error("colour image: expected 8 bits per sample");
  dvo::compat::ImageMat out = dvo::compat::image_create(png.height, png.width);
  float* o = dvo::compat::image_ptr_mut(out);
  for (int y = 0; y < png.height; ++y)
    for (int x = 0; x < png.width; ++x)
      o[size_t(y) * png.width + x] = png.channels >= 3 ? float(greyFromRgb8(png.sample(x, y, 0), png.sample(x, y, 1), png.sample(x, y, 2)))
                                                       : float(png.sample(x, y, 0));
  return out;
}

// SurfacePyramid::convertRawDepthImageSse (surface_pyramid.cpp:65-105): metres = raw * scale, raw 0 -> NaN
inline dvo::compat::ImageMat depthFloatFromPng(const PngImage& png, float scale) {
  if (png.bit_depth != 16 || png.channels != 1) throw std::runtime_error("depth image: expected one 16-bit channel");
  dvo::compat::ImageMat out = dvo::compat::image_create(png.height, png.width);
  float* o = dvo::compat::image_ptr_mut(out);
  for (int y = 0; y < png.height; ++y)
    for (int x = 0; x < png.width; ++x) {
      const unsigned raw = png.sample(x, y, 0);
      o[size_t(y) * png.width + x] = raw == 0 ? std::numeric_limits<float>::quiet_NaN() : float(raw) * scale;
    }
  return out;
}

}  // namespace dvo_benchmark
