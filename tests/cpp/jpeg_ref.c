/* jpeg_ref.c -- TEST INFRASTRUCTURE: the system's libjpeg (libjpeg.so.8 = libjpeg-turbo with the v8 ABI; the image has the
 * library but not its headers), driven the way cv::imdecode(buf, IMREAD_GRAYSCALE) drives it for a JPEG payload
 * (slam_frontend_main.cc:99-100 -> OpenCV 3.2 modules/imgcodecs/src/grfmt_jpeg.cpp, JpegDecoder::readHeader / readData):
 * memory source, jpeg_read_header, out_color_space = JCS_GRAYSCALE, jpeg_start_decompress, one jpeg_read_scanlines per row,
 * jpeg_finish_decompress; a fatal libjpeg error (error_exit -> longjmp) makes imdecode return an empty Mat, warnings
 * (corrupt data, premature end) do not.  Bound by hand: the public prefix of struct jpeg_decompress_struct up to
 * out_color_components is the same in every libjpeg since 6b; its total size is asked of the library (a deliberately wrong
 * size makes jpeg_CreateDecompress report the right one), and the layout is checked by decoding known files.
 *     gcc -O2 -shared -fPIC tests/cpp/jpeg_ref.c -o <out>.so -l:libjpeg.so.8 */
#include <setjmp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern void* jpeg_std_error(void* err);
extern void jpeg_CreateDecompress(void* cinfo, int version, size_t structsize);
extern void jpeg_destroy_decompress(void* cinfo);
extern void jpeg_mem_src(void* cinfo, const unsigned char* buf, unsigned long size);
extern int jpeg_read_header(void* cinfo, int require_image);
extern int jpeg_start_decompress(void* cinfo);
extern unsigned int jpeg_read_scanlines(void* cinfo, unsigned char** rows, unsigned int max_lines);
extern int jpeg_finish_decompress(void* cinfo);

enum { kImageWidth = 48, kImageHeight = 52, kNumComponents = 56, kOutColorSpace = 64, kOutputWidth = 136, kOutputHeight = 140,
       kOutColorComponents = 144, kErrMsgCode = 40, kErrMsgParm = 44, kErrNumWarnings = 128, kErrEmitMessage = 8, kErrOutputMessage = 16 };

static jmp_buf jump;
static int last_code, probed_size;
static long last_warnings;
static void on_error(void* cinfo) {
  char* err = *(char**)cinfo;
  last_code = *(int*)(err + kErrMsgCode);
  probed_size = *(int*)(err + kErrMsgParm);
  longjmp(jump, 1);
}
static void on_output(void* cinfo) { (void)cinfo; }  /* (no text on stderr) */

int jpeg_ref_last_error_code(void) { return last_code; }
long jpeg_ref_warnings(void) { return last_warnings; }

/* -> 0: decoded (out: rows of `pitch` bytes; *w, *h set); 1: the header was refused; 2: the data was refused (a fatal error
 * while decoding: cv::imdecode returns an empty Mat); 3: another size than (want_w, want_h) when both are > 0; 4: a file of
 * four components (OpenCV converts CMYK itself: not followed here). */
int jpeg_ref_gray(const unsigned char* data, size_t size, int want_w, int want_h, unsigned char* out, size_t pitch, int* w_out,
                  int* h_out) {
  static char err[1024];
  static size_t struct_size;
  char* volatile cinfo = NULL;
  unsigned char* volatile row = NULL;
  volatile int stage = 1;
  last_code = 0;
  last_warnings = 0;
  memset(err, 0, sizeof(err));
  jpeg_std_error(err);
  *(void**)err = (void*)on_error;
  *(void**)(err + kErrOutputMessage) = (void*)on_output;
  if (struct_size == 0) {  /* ask the library how large its struct is */
    char* probe = (char*)calloc(1, 8192);
    *(void**)probe = err;
    if (!setjmp(jump)) jpeg_CreateDecompress(probe, 80, 12345);
    free(probe);
    if (probed_size < 400 || probed_size > 4096) return 1;
    struct_size = (size_t)probed_size;
  }
  cinfo = (char*)calloc(1, struct_size + 64);
  *(void**)cinfo = err;
  if (setjmp(jump)) {
    last_warnings = *(long*)(err + kErrNumWarnings);
    jpeg_destroy_decompress(cinfo);
    free(cinfo);
    free(row);
    return stage;
  }
  jpeg_CreateDecompress(cinfo, 80, struct_size);
  jpeg_mem_src(cinfo, data, (unsigned long)size);
  jpeg_read_header(cinfo, 1);
  const int w = *(int*)(cinfo + kImageWidth), h = *(int*)(cinfo + kImageHeight);
  if (w_out) *w_out = w;
  if (h_out) *h_out = h;
  if (want_w > 0 && want_h > 0 && (w != want_w || h != want_h)) {
    jpeg_destroy_decompress(cinfo);
    free(cinfo);
    return 3;
  }
  if (*(int*)(cinfo + kNumComponents) == 4) {
    jpeg_destroy_decompress(cinfo);
    free(cinfo);
    return 4;
  }
  stage = 2;
  *(int*)(cinfo + kOutColorSpace) = 1;       /* JCS_GRAYSCALE */
  *(int*)(cinfo + kOutColorComponents) = 1;
  jpeg_start_decompress(cinfo);
  row = (unsigned char*)malloc((size_t)w * 4 + 64);
  for (int y = 0; y < h; y++) {
    unsigned char* rows[1] = {row};
    jpeg_read_scanlines(cinfo, rows, 1);
    memcpy(out + (size_t)y * pitch, row, (size_t)w);
  }
  jpeg_finish_decompress(cinfo);
  last_warnings = *(long*)(err + kErrNumWarnings);
  jpeg_destroy_decompress(cinfo);
  free(cinfo);
  free(row);
  return 0;
}
