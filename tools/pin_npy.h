// pin_npy.h -- the smallest possible reader / writer of NumPy .npy files (format version 1.0, little endian, C order),
// shared by tools/pin_with_opencv.cc (the OpenCV pin kit) and tests/cpp/test_pin_npy.cc (which proves, without OpenCV, that
// what this header writes is what numpy.load reads, and the reverse).  No dependency beyond the C++ standard library.
#ifndef VSF_PIN_NPY_H_
#define VSF_PIN_NPY_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pin_npy {

struct Array {
  std::string descr;           // "|u1", "<i4", "<f4", or a structured list like "[('x', '<f4'), ...]"
  std::vector<size_t> shape;   // C order
  size_t itemsize = 0;         // bytes per element (given by the caller when writing, derived from descr when reading)
  std::vector<uint8_t> data;
  size_t count() const {
    size_t n = 1;
    for (size_t s : shape) n *= s;
    return n;
  }
};

// cv::KeyPoint / cv::DMatch as numpy sees them (oracle/binding.py: KEYPOINT_DTYPE, DMATCH_DTYPE)
static const char* const kKeyPointDescr =
    "[('x', '<f4'), ('y', '<f4'), ('size', '<f4'), ('angle', '<f4'), ('response', '<f4'), ('octave', '<i4'), ('class_id', '<i4')]";
static const char* const kDMatchDescr = "[('queryIdx', '<i4'), ('trainIdx', '<i4'), ('imgIdx', '<i4'), ('distance', '<f4')]";

inline void write(const std::string& path, const std::string& descr, const std::vector<size_t>& shape, const void* data,
                  size_t nbytes) {
  std::string d = descr;
  if (d.empty() || d[0] != '[') d = "'" + d + "'";
  std::string shp = "(";
  for (size_t i = 0; i < shape.size(); i++) shp += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  shp += ")";
  std::string h = "{'descr': " + d + ", 'fortran_order': False, 'shape': " + shp + ", }";
  const size_t unpadded = 10 + h.size() + 1;  // magic (6) + version (2) + header length (2) + header + '\n'
  h.append((64 - unpadded % 64) % 64, ' ');
  h.push_back('\n');
  if (h.size() > 65535) throw std::runtime_error("npy header too long: " + path);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  const unsigned char len[2] = {(unsigned char)(h.size() & 255), (unsigned char)(h.size() >> 8)};
  bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(len, 1, 2, f) == 2 && std::fwrite(h.data(), 1, h.size(), f) == h.size() &&
            (nbytes == 0 || std::fwrite(data, 1, nbytes, f) == nbytes);
  ok = std::fclose(f) == 0 && ok;
  if (!ok) throw std::runtime_error("short write: " + path);
}

inline size_t itemsize_of(const std::string& descr) {
  // plain: one type code; structured: the sum of the '<f4'-style codes inside
  size_t total = 0;
  for (size_t i = 0; i + 2 < descr.size(); i++)
    if ((descr[i] == '<' || descr[i] == '|' || descr[i] == '=') && std::strchr("uifb", descr[i + 1]) && descr[i + 2] >= '1' &&
        descr[i + 2] <= '8')
      total += (size_t)(descr[i + 2] - '0');
  return total;
}

inline Array read(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot read " + path);
  unsigned char head[10];
  if (std::fread(head, 1, 10, f) != 10 || std::memcmp(head, "\x93NUMPY", 6) != 0 || head[6] != 1) {
    std::fclose(f);
    throw std::runtime_error("not a version-1 .npy file: " + path);
  }
  const size_t hlen = head[8] | ((size_t)head[9] << 8);
  std::string h(hlen, '\0');
  if (std::fread(&h[0], 1, hlen, f) != hlen) {
    std::fclose(f);
    throw std::runtime_error("truncated header: " + path);
  }
  Array a;
  const size_t dk = h.find("'descr':");
  const size_t fk = h.find("'fortran_order':");
  const size_t sk = h.find("'shape':");
  if (dk == std::string::npos || fk == std::string::npos || sk == std::string::npos || h.find("True", fk) < sk) {
    std::fclose(f);
    throw std::runtime_error("unsupported header (Fortran order?): " + path);
  }
  size_t d0 = h.find_first_not_of(' ', dk + 8);
  if (h[d0] == '[') {
    a.descr = h.substr(d0, h.find(']', d0) - d0 + 1);
  } else {
    const size_t q = h.find('\'', d0 + 1);
    a.descr = h.substr(d0 + 1, q - d0 - 1);
  }
  const size_t p0 = h.find('(', sk), p1 = h.find(')', p0);
  std::string dims = h.substr(p0 + 1, p1 - p0 - 1);
  size_t pos = 0;
  while (pos < dims.size()) {
    while (pos < dims.size() && (dims[pos] == ' ' || dims[pos] == ',')) pos++;
    if (pos >= dims.size()) break;
    a.shape.push_back((size_t)std::stoull(dims.substr(pos)));
    while (pos < dims.size() && dims[pos] != ',') pos++;
  }
  a.itemsize = itemsize_of(a.descr);
  a.data.resize(a.count() * a.itemsize);
  const bool ok = a.data.empty() || std::fread(a.data.data(), 1, a.data.size(), f) == a.data.size();
  std::fclose(f);
  if (!ok || a.itemsize == 0) throw std::runtime_error("truncated data: " + path);
  return a;
}

}  // namespace pin_npy
#endif  // VSF_PIN_NPY_H_
