/* png_ref.c -- TEST INFRASTRUCTURE: the real libpng (the system's libpng16.so.16, bound by hand: the image has the library
 * but not its headers), driven the way cv::imdecode(buf, IMREAD_GRAYSCALE) drives it for a PNG payload
 * (slam_frontend_main.cc:99-100 -> OpenCV 3.2 modules/imgcodecs/src/grfmt_png.cpp, PngDecoder::readHeader / readData with
 * an 8-bit one-channel destination): read callback over the buffer, png_read_info, strip_16 for 16-bit files, strip_alpha,
 * palette_to_rgb, expand_gray_1_2_4_to_8, rgb_to_gray(1, 0.299, 0.587), interlace handling, png_read_image, png_read_end;
 * any png_error on the way (longjmp) means imdecode returns an empty Mat.  The parity tests compare
 * vsf_png_decode_gray_batch with this -- bytes and refusals.
 *     gcc -O2 -shared -fPIC tests/cpp/png_ref.c -o <out>.so -l:libpng16.so.16 */
#include <setjmp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void* png_structp;
typedef void* png_infop;
extern png_structp png_create_read_struct(const char* ver, void* error_ptr, void (*error_fn)(png_structp, const char*),
                                          void (*warn_fn)(png_structp, const char*));
extern png_infop png_create_info_struct(png_structp);
extern void png_destroy_read_struct(png_structp*, png_infop*, png_infop*);
extern jmp_buf* png_set_longjmp_fn(png_structp, void (*)(jmp_buf, int), size_t);
extern void png_set_read_fn(png_structp, void* io_ptr, void (*read_fn)(png_structp, unsigned char*, size_t));
extern void* png_get_io_ptr(png_structp);
extern void png_read_info(png_structp, png_infop);
extern uint32_t png_get_IHDR(png_structp, png_infop, uint32_t* w, uint32_t* h, int* depth, int* color, int* interlace, int* comp,
                             int* filter);
extern void png_set_strip_16(png_structp);
extern void png_set_strip_alpha(png_structp);
extern void png_set_palette_to_rgb(png_structp);
extern void png_set_expand_gray_1_2_4_to_8(png_structp);
extern void png_set_rgb_to_gray(png_structp, int error_action, double red, double green);
extern int png_set_interlace_handling(png_structp);
extern void png_read_update_info(png_structp, png_infop);
extern void png_read_image(png_structp, unsigned char** rows);
extern void png_read_end(png_structp, png_infop);
extern void png_error(png_structp, const char*);
extern const char* png_get_libpng_ver(png_structp);
extern size_t png_get_rowbytes(png_structp, png_infop);

typedef struct {
  const unsigned char* data;
  size_t size, pos;
} Source;

static void read_from_buffer(png_structp png, unsigned char* dst, size_t n) {
  Source* s = (Source*)png_get_io_ptr(png);
  if (s->pos + n > s->size) png_error(png, "PNG input buffer is incomplete");
  memcpy(dst, s->data + s->pos, n);
  s->pos += n;
}
extern void png_longjmp(png_structp, int);
static char last_error[256], last_warning[256];
static void quiet(png_structp png, const char* msg) {  /* warnings (and the benign errors libpng turns into them on read) */
  (void)png;
  strncpy(last_warning, msg ? msg : "", sizeof(last_warning) - 1);
}
static void failed(png_structp png, const char* msg) {
  strncpy(last_error, msg ? msg : "", sizeof(last_error) - 1);
  png_longjmp(png, 1);
}
const char* png_ref_last_error(void) { return last_error; }
const char* png_ref_last_warning(void) { return last_warning; }

const char* png_ref_version(void) { return png_get_libpng_ver(NULL); }

/* -> 0: decoded (out: h rows of `pitch` bytes, w used; *w, *h set); 1: the header was refused; 2: the data was refused (a
 * png_error in readData: cv::imdecode returns an empty image); 3: the file's size is not (want_w, want_h) (when both > 0).
 * info[0..3]: bit depth, colour type, interlace, rows' byte count after the transformations. */
int png_ref_gray(const unsigned char* data, size_t size, int want_w, int want_h, unsigned char* out, size_t pitch, int* w_out,
                 int* h_out, int* info) {
  Source src = {data, size, 0};
  png_structp png = png_create_read_struct(png_get_libpng_ver(NULL), NULL, failed, quiet);
  last_error[0] = last_warning[0] = 0;
  if (!png) return 1;
  png_infop inf = png_create_info_struct(png), end = png_create_info_struct(png);
  unsigned char** volatile rows = NULL;
  volatile int stage = 1;
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
  if (setjmp(*png_set_longjmp_fn(png, longjmp, sizeof(jmp_buf)))) {
    png_destroy_read_struct(&png, &inf, &end);
    free((void*)rows);
    return stage;
  }
  png_set_read_fn(png, &src, read_from_buffer);
  png_read_info(png, inf);
  png_get_IHDR(png, inf, &w, &h, &depth, &color, &interlace, NULL, NULL);
  if (!(depth <= 8 || depth == 16)) {
    png_destroy_read_struct(&png, &inf, &end);
    return 1;
  }
  if (w_out) *w_out = (int)w;
  if (h_out) *h_out = (int)h;
  if (info) {
    info[0] = depth;
    info[1] = color;
    info[2] = interlace;
  }
  if (want_w > 0 && want_h > 0 && ((int)w != want_w || (int)h != want_h)) {
    png_destroy_read_struct(&png, &inf, &end);
    return 3;
  }
  stage = 2;
  if (depth == 16) png_set_strip_16(png);
  png_set_strip_alpha(png);
  if (color == 3) png_set_palette_to_rgb(png);
  if ((color & 2) == 0 && depth < 8) png_set_expand_gray_1_2_4_to_8(png);
  png_set_rgb_to_gray(png, 1, 0.299, 0.587);
  png_set_interlace_handling(png);
  png_read_update_info(png, inf);
  if (info) info[3] = (int)png_get_rowbytes(png, inf);
  if (png_get_rowbytes(png, inf) > pitch) png_error(png, "row does not fit");
  rows = (unsigned char**)malloc(sizeof(unsigned char*) * (h ? h : 1));
  for (uint32_t y = 0; y < h; y++) rows[y] = out + (size_t)y * pitch;
  png_read_image(png, (unsigned char**)rows);
  png_read_end(png, end);
  png_destroy_read_struct(&png, &inf, &end);
  free((void*)rows);
  return 0;
}
