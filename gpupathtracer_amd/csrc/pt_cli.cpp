// Command-line host: the reference's main() (src/ptMain.cu:500-698) over the C ABI.
//
// Same flags and defaults as the reference (src/ptMain.cu:502-549): -q/--quick, -c/--cpu,
// -m/--median, -w/--width, -h/--height (height, not help), -n/--numsamples, -t/--threads,
// -d/--maxdepth, -s/--stacksize, -f/--file; plus --scene, --gpus, --accel, --pipeline, which the
// reference lacks (its scene is a source edit at src/ptMain.cu:559), and --stream-in / --stream-out,
// which read and write the scene byte stream main() builds for its device (src/ptMain.cu:560-572),
// and --pass-samples K, which renders in passes of K samples and reports progress after each
// (the reference's Progress display, src/ptMain.cu:663-677, on the GPU path), and --earthmap, the texels of
// the final scene's earth sphere (the reference decodes earthmap.jpg with stb_image in its host branch,
// src/ptMain.cu:396-402; here a binary PPM, or "synthetic" for a fixed 512x256 pattern).  The reference parses with
// the vendored cxxopts; getopt_long gives the same short/long spellings here.
#include <getopt.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_image_io.h"

namespace {

int fail(const char* what)
{
    std::cerr << what << ": " << pt_last_error() << std::endl;
    return EXIT_FAILURE;
}

// Texels for the final scene's earth sphere: a baseline JPEG (what the reference loads), a binary PPM (P6, maxval 255), or the word "synthetic" for a
// fixed 512x256 pattern (the size of the reference's earthmap.jpg; not its content).
bool load_earthmap(const std::string& arg, std::vector<uint8_t>* texels, int* w, int* h)
{
    if (arg == "synthetic") {
        *w = 512; *h = 256;
        texels->resize((size_t)3 * 512 * 256);
        for (int j = 0; j < 256; j++)
            for (int i = 0; i < 512; i++) {
                uint8_t* t = &(*texels)[3 * ((size_t)j * 512 + i)];
                t[0] = (uint8_t)((i * 7 + j * 13) & 255); t[1] = (uint8_t)((i ^ (j * 3)) & 255); t[2] = (uint8_t)(((i * j) >> 3) & 255);
            }
        return true;
    }
    {
        // a JPEG file, as the reference reads (stbi_load("earthmap.jpg", ...), src/ptMain.cu:399): decoded by the library
        std::ifstream jf(arg, std::ios::binary);
        unsigned char head[2] = { 0, 0 };
        if (jf && jf.read(reinterpret_cast<char*>(head), 2) && head[0] == 0xFF && head[1] == 0xD8) {
            jf.seekg(0, std::ios::end);
            std::vector<char> file((size_t)jf.tellg());
            jf.seekg(0);
            jf.read(file.data(), (std::streamsize)file.size());
            int32_t jw = 0, jh = 0;
            if (pt_image_decode_jpeg(file.data(), file.size(), &jw, &jh, nullptr, 0) != PT_OK) { std::cerr << "--earthmap: " << arg << ": " << pt_last_error() << std::endl; return false; }
            texels->resize((size_t)3 * jw * jh);
            if (pt_image_decode_jpeg(file.data(), file.size(), &jw, &jh, texels->data(), texels->size()) != PT_OK) { std::cerr << "--earthmap: " << pt_last_error() << std::endl; return false; }
            *w = jw; *h = jh;
            return true;
        }
    }
    std::ifstream f(arg, std::ios::binary);
    std::string magic;
    int maxval = 0;
    if (!f || !(f >> magic >> *w >> *h >> maxval) || magic != "P6" || maxval != 255 || *w <= 0 || *h <= 0 || (long)*w * *h > (1L << 26)) {
        std::cerr << "--earthmap: " << arg << " is neither a JPEG nor a binary PPM (P6, maxval 255)" << std::endl;
        return false;
    }
    f.get();                                     // the single whitespace byte after maxval
    texels->resize((size_t)3 * *w * *h);
    f.read(reinterpret_cast<char*>(texels->data()), (std::streamsize)texels->size());
    if (f.gcount() != (std::streamsize)texels->size()) { std::cerr << "--earthmap: " << arg << " is truncated" << std::endl; return false; }
    return true;
}

}  // namespace

int main(int argc, char** argv)
{
    bool quick = false, cpu = false, median = false;
    int ns = 100, nx = 128 * 4, ny = 128 * 4, maxDepth = 25, gpus = 1, passSamples = 0;
    std::string outFile = "outputImage.ppm", scene = "random", accel = "auto", pipeline = "fused", streamIn, streamOut, earthmap;

    static const option longopts[] = {
        { "quick", no_argument, nullptr, 'q' }, { "cpu", no_argument, nullptr, 'c' }, { "median", no_argument, nullptr, 'm' },
        { "width", required_argument, nullptr, 'w' }, { "height", required_argument, nullptr, 'h' },
        { "numsamples", required_argument, nullptr, 'n' }, { "threads", required_argument, nullptr, 't' },
        { "maxdepth", required_argument, nullptr, 'd' }, { "stacksize", required_argument, nullptr, 's' },
        { "file", required_argument, nullptr, 'f' }, { "scene", required_argument, nullptr, 1000 },
        { "gpus", required_argument, nullptr, 1001 }, { "accel", required_argument, nullptr, 1002 },
        { "pipeline", required_argument, nullptr, 1003 }, { "stream-in", required_argument, nullptr, 1004 },
        { "stream-out", required_argument, nullptr, 1005 }, { "pass-samples", required_argument, nullptr, 1006 },
        { "earthmap", required_argument, nullptr, 1007 },
        { nullptr, 0, nullptr, 0 } };
    int c;
    while ((c = getopt_long(argc, argv, "qcmw:h:n:t:d:s:f:", longopts, nullptr)) != -1) {
        switch (c) {
        case 'q': quick = true; break;
        case 'c': cpu = true; break;
        case 'm': median = true; break;
        case 'w': nx = std::atoi(optarg); break;
        case 'h': ny = std::atoi(optarg); break;
        case 'n': ns = std::atoi(optarg); break;
        case 't': break;                         // render threads: host-loop option, nothing to do on the GPU path
        case 'd': maxDepth = std::atoi(optarg); break;
        case 's': break;                         // GPU thread stack size: the kernels do not recurse
        case 'f': outFile = optarg; break;
        case 1000: scene = optarg; break;
        case 1001: gpus = std::atoi(optarg); break;
        case 1002: accel = optarg; break;
        case 1003: pipeline = optarg; break;
        case 1004: streamIn = optarg; break;
        case 1005: streamOut = optarg; break;
        case 1006: passSamples = std::atoi(optarg); break;
        case 1007: earthmap = optarg; break;
        default:
            std::cerr << "usage: ptrace [-q] [-w W] [-h H] [-n SPP] [-d DEPTH] [-f out.{ppm,hdr,png,tga,bmp}]\n"
                         "              [--scene simple_spheres|simple_light|random|cornell|cornell_spheres|final|trimesh]\n"
                         "              [--gpus N] [--accel auto|list|lbvh] [--pipeline fused|wavefront]\n"
                         "              [--stream-in scene.bin] [--stream-out scene.bin] [--pass-samples K] [-m]\n"
                         "              [--earthmap map.jpg|map.ppm|synthetic]   (scene final: texels of the earth sphere)" << std::endl;
            return EXIT_FAILURE;
        }
    }
    if (cpu) { std::cerr << "-c/--cpu: this build renders on the GPU only; it has no CPU path." << std::endl; return EXIT_FAILURE; }
    if (quick) { nx /= 8; ny /= 8; ns /= 16; }   // src/ptMain.cu:544-549
    if (nx <= 0 || ny <= 0 || ns <= 0 || gpus <= 0) { std::cerr << "width, height, numsamples and gpus must be positive" << std::endl; return EXIT_FAILURE; }

    const float aspect = float(nx) / float(ny);
    PtSceneDesc* desc = nullptr;
    if (!streamIn.empty()) {                     // a scene the reference serialised (its camera fixes the aspect)
        std::ifstream f(streamIn, std::ios::binary);
        if (!f) { std::cerr << "cannot open " << streamIn << std::endl; return EXIT_FAILURE; }
        const std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (pt_scene_desc_from_stream(blob.data(), blob.size(), 0, &desc) != PT_OK) return fail("scene stream");
    } else {
        std::vector<uint8_t> texels;
        int tw = 0, th = 0;
        if (!earthmap.empty() && !load_earthmap(earthmap, &texels, &tw, &th)) return EXIT_FAILURE;
        if (pt_scene_desc_builtin(scene.c_str(), aspect, texels.empty() ? nullptr : texels.data(), tw, th, &desc) != PT_OK) return fail("scene");
    }
    if (!streamOut.empty()) {
        size_t size = 0;
        if (pt_scene_desc_to_stream(desc, nullptr, 0, &size, nullptr) != PT_OK) return fail("scene stream");
        std::vector<char> blob(size);
        if (pt_scene_desc_to_stream(desc, blob.data(), blob.size(), &size, nullptr) != PT_OK) return fail("scene stream");
        std::ofstream f(streamOut, std::ios::binary);
        f.write(blob.data(), (std::streamsize)blob.size());
        if (!f) { std::cerr << "cannot write " << streamOut << std::endl; return EXIT_FAILURE; }
    }
    if (accel == "lbvh") desc->accel = PT_ACCEL_LBVH;
    else if (accel == "list") desc->accel = PT_ACCEL_LIST;
    else if (desc->n_world > 32 || desc->n_bvhs > 0) desc->accel = PT_ACCEL_LBVH;      // the library falls back to the list where its tree cannot stand for it

    const int have = pt_device_count();
    if (have < 1) { std::cerr << "no HIP device is visible" << std::endl; return EXIT_FAILURE; }
    if (gpus > have) { std::cerr << "--gpus " << gpus << " but only " << have << " device(s) visible" << std::endl; return EXIT_FAILURE; }

    PtRenderParams p{};
    p.nx = nx; p.ny = ny; p.ns = ns; p.max_depth = maxDepth; p.tile_w = 8; p.tile_h = 8;
    p.pipeline = (pipeline == "wavefront") ? PT_PIPELINE_WAVEFRONT : PT_PIPELINE_FUSED;

    if (passSamples > 0 && gpus == 1) {          // one device, progress after every pass
        PtScene* sc = nullptr;
        std::cerr << "Allocating world...";
        if (pt_set_device(0) != PT_OK || pt_scene_create(desc, &sc) != PT_OK) return fail("\nFailed to allocate world in GPU memory");
        std::cerr << "done" << std::endl;
        std::vector<float> image((size_t)3 * nx * ny);
        auto report = [](int32_t done, int32_t total, void*) -> int {
            std::cerr << "\rRendering world... " << (100 * (long)done / total) << "% (" << done << "/" << total << " samples)" << std::flush;
            return 0;
        };
        if (pt_render_progressive(sc, &p, image.data(), passSamples, report, nullptr) != PT_OK) return fail("\nFailed to render on GPU");
        std::cerr << std::endl;
        pt_scene_destroy(sc);
        pt_scene_desc_free(desc);
        if (median) {
            std::vector<float> filtered(image.size());
            if (pt_median_filter(image.data(), filtered.data(), nx, ny, 3) != PT_OK) return fail("median filter");
            image.swap(filtered);
        }
        std::string err;
        if (!pt::write_image(outFile, image.data(), nx, ny, &err)) { std::cerr << err << std::endl; return EXIT_FAILURE; }
        std::cerr << "Done." << std::endl;
        return EXIT_SUCCESS;
    }
    if (passSamples > 0) std::cerr << "--pass-samples applies to one GPU; rendering in one pass per device." << std::endl;

    std::cerr << "Allocating and rendering world...";
    const size_t n_floats = (size_t)3 * nx * ny;
    std::vector<float> image(n_floats, 0.0f);
    std::vector<int32_t> ids((size_t)gpus);
    for (int g = 0; g < gpus; g++) ids[(size_t)g] = g;
    const auto t0 = std::chrono::steady_clock::now();
    if (pt_render_on_devices(desc, &p, ids.data(), gpus, image.data()) != PT_OK) return fail("\nFailed to render on GPU");
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "done" << std::endl;
    std::cerr << (double)nx * ny * ns / sec / 1e6 << " Mpaths/s including scene upload (" << sec << " s, " << gpus << " GPU)" << std::endl;
    pt_scene_desc_free(desc);
    if (median) {                                // medianFilter3x3 before writeImage, src/ptMain.cu:684-689
        std::vector<float> filtered(n_floats);
        pt_set_device(0);
        if (pt_median_filter(image.data(), filtered.data(), nx, ny, 3) != PT_OK) return fail("median filter");
        image.swap(filtered);
    }

    std::string err;
    if (!pt::write_image(outFile, image.data(), nx, ny, &err)) { std::cerr << err << std::endl; return EXIT_FAILURE; }
    std::cerr << "Done." << std::endl;
    return EXIT_SUCCESS;
}
