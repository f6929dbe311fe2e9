// image.cpp -- image egress: the output side of `render(ply, camera) -> image`
// (app/main.cpp:310-340): CHW float -> vertically flipped HWC uint8 with a truncating `* 255`, then an
// 8-bit RGB PNG (the reference calls stbi_write_png(name, w, h, 3, data, 0); stb is not part of the
// reference tree, so the PNG container is written here with stored deflate blocks).
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../common.hpp"

namespace
{

uint32_t crc_table[256];
bool     crc_ready = false;
void     crc_init()
{
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
    if (!crc_ready) crc_init();
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void write_chunk(FILE* fp, const char type[4], const std::vector<uint8_t>& data)
{
    std::vector<uint8_t> hdr;
    put_be32(hdr, (uint32_t)data.size());
    fwrite(hdr.data(), 1, 4, fp);
    fwrite(type, 1, 4, fp);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), fp);
    uint32_t crc = 0xFFFFFFFFu;
    crc          = crc32_update(crc, reinterpret_cast<const uint8_t*>(type), 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    crc ^= 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    fwrite(tail.data(), 1, 4, fp);
}

// CHW float -> flipped HWC u8 (app/main.cpp:323-335); one lane per output pixel, coalesced 3-plane reads
__global__ void k_image_to_rgb8(int w, int h, const float* __restrict__ img, uint8_t* __restrict__ rgb)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)w * h) return;
    const int     i   = (int)(p / w), j = (int)(p % w);
    const int64_t idx = (int64_t)(h - i - 1) * w + j;
    const int64_t hw  = (int64_t)w * h;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = (uint8_t)(int)(img[c * hw + idx] * 255.0f);
}

// mean squared error of two images and its gradient (2 (a - b) / n); the partial sums of a workgroup go out as one atomic
__global__ void __launch_bounds__(256) k_l2_loss_backward(int64_t n, const float* __restrict__ img,
                                                          const float* __restrict__ target, float* __restrict__ dL,
                                                          float* __restrict__ loss)
{
    __shared__ float s_w[4];
    const float      inv = 1.0f / (float)n;
    float            acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = img[i] - target[i];
        dL[i]         = 2.0f * d * inv;
        acc += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (s_w[0] + s_w[1] + s_w[2] + s_w[3]) * inv);
}

} // namespace

extern "C" {

lcgs_status lcgs_l2_loss_backward(lcgs_context* ctx, int width, int height, const float* d_img_chw,
                                  const float* d_target_chw, float* d_dL_dimg, float* d_loss)
{
    if (!ctx || !d_img_chw || !d_target_chw || !d_dL_dimg || !d_loss || width <= 0 || height <= 0) {
        lcgs::set_last_error("lcgs_l2_loss_backward: invalid argument");
        return LCGS_ERR_INVALID_ARG;
    }
    const int64_t n  = (int64_t)width * height * 3;
    hipStream_t   st = lcgs::context_stream(ctx);
    hipError_t    e  = hipSetDevice(lcgs::context_device(ctx)); // multi-GPU processes: every entry point selects its device
    if (e != hipSuccess) return lcgs::hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    e = hipMemsetAsync(d_loss, 0, sizeof(float), st);
    if (e != hipSuccess) return lcgs::hip_fail(e, "hipMemsetAsync(loss)", __FILE__, __LINE__);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_l2_loss_backward, dim3((unsigned)blocks), dim3(256), 0, st, n, d_img_chw, d_target_chw, d_dL_dimg,
                       d_loss);
    e = hipGetLastError();
    if (e != hipSuccess) return lcgs::hip_fail(e, "k_l2_loss_backward", __FILE__, __LINE__);
    return LCGS_OK;
}

void lcgs_image_to_rgb8(int width, int height, const float* h_img_chw, uint8_t* h_rgb)
{
    const int w = width, h = height;
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            const int pixel_idx = (i * w + j) * 3;
            const int idx       = (h - i - 1) * w + j; // vertical flip, app/main.cpp:331
            for (int c = 0; c < 3; ++c)
                h_rgb[pixel_idx + c] = (uint8_t)(int)(h_img_chw[(size_t)c * h * w + idx] * 255.0f);
        }
    }
}

lcgs_status lcgs_image_to_rgb8_device(lcgs_context* ctx, int width, int height, const float* d_img_chw, uint8_t* d_rgb)
{
    if (!ctx || !d_img_chw || !d_rgb || width <= 0 || height <= 0) {
        lcgs::set_last_error("lcgs_image_to_rgb8_device: invalid argument");
        return LCGS_ERR_INVALID_ARG;
    }
    const int64_t n = (int64_t)width * height;
    hipLaunchKernelGGL(k_image_to_rgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lcgs::context_stream(ctx), width,
                       height, d_img_chw, d_rgb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lcgs::hip_fail(e, "k_image_to_rgb8", __FILE__, __LINE__);
    return LCGS_OK;
}

lcgs_status lcgs_write_png(const char* path, int width, int height, const uint8_t* h_rgb)
{
    if (!path || !h_rgb || width <= 0 || height <= 0) {
        lcgs::set_last_error("lcgs_write_png: invalid argument");
        return LCGS_ERR_INVALID_ARG;
    }
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        lcgs::set_last_error(std::string("cannot open ") + path + " for writing");
        return LCGS_ERR_IO;
    }
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    fwrite(sig, 1, 8, fp);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(2); // colour type RGB
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    write_chunk(fp, "IHDR", ihdr);

    // raw scanlines (filter byte 0 + RGB row), wrapped in a zlib stream of stored blocks
    const size_t         row = (size_t)width * 3 + 1;
    std::vector<uint8_t> rawdata(row * height);
    for (int y = 0; y < height; ++y) {
        rawdata[y * row] = 0;
        memcpy(&rawdata[y * row + 1], h_rgb + (size_t)y * width * 3, (size_t)width * 3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t   pos = 0;
    while (pos < rawdata.size()) {
        size_t n = std::min<size_t>(65535, rawdata.size() - pos);
        z.push_back(pos + n == rawdata.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), rawdata.begin() + pos, rawdata.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) {
            a = (a + rawdata[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    write_chunk(fp, "IDAT", z);
    write_chunk(fp, "IEND", {});
    fclose(fp);
    return LCGS_OK;
}

} // extern "C"
