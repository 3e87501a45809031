// wm_app.cpp -- the reference's sample application protocol (Watermark_GPU/main.cpp) on the MI355X engine.
//
// Reads the reference's settings.ini unchanged (same sections/keys, settings.ini:1-25; `opencl_device` is taken as
// the HIP device ordinal) and runs either
//   * testForImage (main.cpp:140-242): load RGB image, grey = 0.299R+0.587G+0.114B, warm-up, `loops_for_test` timed
//     NVF and ME embeds on the RGB base, detects on the grey of the watermarked images, prints strength, FPS and the
//     two correlations in the reference's format, optionally saves <name>_W_NVF / <name>_W_ME, or
//   * testForVideo (main.cpp:245-410): raw yuv420p frames (a .y4m file, or a headerless .yuv with the optional keys
//     [parameters_video] video_width / video_height) -- every `watermark_interval`-th frame gets its Y plane
//     watermarked (ME mask) and Y'UV frames are written in order to `encode_watermark_file_path` (.y4m/.yuv, or "-"
//     for stdout so that an external `ffmpeg -f rawvideo ...` can encode as the reference's pipe does), or, when
//     `watermark_detection = true`, "Correlation for frame: i: c" is printed per interval frame.
// FFmpeg demux/decode/encode (main.cpp:255-263,284-293) is out of scope: containers are decoded outside.
// Frames are staged from pinned host buffers through the engine's slots, several frames in flight.
#include "../../../include/Watermark.hpp"
#include "imageio.hpp"
#include "ini.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

using std::cout;
using std::string;

namespace timer {  // Utilities.cpp:13-27
static std::chrono::time_point<std::chrono::steady_clock> startTime, currentTime;
static void start() { startTime = std::chrono::steady_clock::now(); }
static void end() { currentTime = std::chrono::steady_clock::now(); }
static float elapsedSeconds() { return (float)(std::chrono::duration_cast<std::chrono::microseconds>(currentTime - startTime).count() / 1000000.0f); }
}  // namespace timer

static string addSuffixBeforeExtension(const string& file, const string& suffix)  // Utilities.cpp:7-11
{
    auto dot = file.find_last_of('.');
    return dot == string::npos ? file + suffix : file.substr(0, dot) + suffix + file.substr(dot);
}

static string executionTime(const bool showFps, const double seconds)  // main.cpp:464-467
{
    char buf[64];
    if (showFps) std::snprintf(buf, sizeof buf, "FPS: %.2f FPS", 1.0 / seconds);
    else std::snprintf(buf, sizeof buf, "%.6f seconds", seconds);
    return buf;
}

static void exitProgram(int code) { std::exit(code); }  // main.cpp:470-474 without the Windows "pause"

static void checkError(bool cond, const string& msg)  // main.cpp:47-54
{
    if (cond) { cout << msg << "\n"; exitProgram(EXIT_FAILURE); }
}

// rgb2gray with the harness weights (main.cpp:142-144,154): f32, (r*R + g*G) + b*B
static std::vector<float> rgb2gray(const std::vector<float>& planar, size_t n)
{
    std::vector<float> g(n);
    for (size_t i = 0; i < n; ++i) g[i] = 0.299f * planar[i] + 0.587f * planar[n + i] + 0.114f * planar[2 * n + i];
    return g;
}

// The image protocol of the reference's sample application (main.cpp:140-242), restructured as a table over the two masks:
// load + grey conversion, one warm-up call per mask, `loops_for_test` timed synchronous calls per operation (average, as
// seconds or FPS), the strengths, the correlations of the grey of the watermarked images, optional saving.  The printed
// lines keep the reference's wording so that its output can be diffed.
namespace {
struct MaskRun {
    MASK_TYPE type;
    const char* name;        // "NVF" / "ME" as the reference prints them
    const char* suffix;      // output file suffix (Utilities.cpp:7-11 naming)
    wm::Image marked;        // makeWatermark's result on the RGB base
    std::vector<float> host; // ... downloaded (planar RGB)
    wm::Image markedGray;    // its grey, what the detector is given (main.cpp:196-197)
    float strength = 0.0f, correlation = 0.0f;
};

template <typename F>
double averageSeconds(int loops, F&& call)
{
    double total = 0.0;
    for (int i = 0; i < loops; ++i) {
        timer::start();
        call();
        timer::end();
        total += timer::elapsedSeconds();
    }
    return total / loops;
}
}  // namespace

static int testForImage(const INIReader& inir, const int p, const float psnr, const int device)
{
    const string imageFile = inir.Get("paths", "image", "NO_IMAGE");
    const bool showFps = inir.GetBoolean("options", "execution_time_in_fps", false);
    const long requested = inir.GetInteger("parameters", "loops_for_test", 5);
    const int loops = requested > 0 ? (int)requested : 5;
    cout << "Each test will be executed " << loops << " times. Average time will be shown below\n";

    timer::start();
    const RgbImage im = read_image(imageFile);
    const dim_t rows = im.rows, cols = im.cols;
    const size_t n = (size_t)rows * cols;
    std::vector<float> planar(3 * n);  // interleaved u8 -> planar f32 (the engine's RGB layout)
    for (int ch = 0; ch < 3; ++ch)
        for (size_t i = 0; i < n; ++i) planar[ch * n + i] = (float)im.rgb[3 * i + ch];
    const wm::Image rgbImage = wm::Image::fromHost(planar.data(), rows, cols, 3, device);
    const wm::Image image = wm::Image::fromHost(rgb2gray(planar, n).data(), rows, cols, 1, device);
    timer::end();
    cout << "Time to load and transfer RGB image from disk to VRAM: " << timer::elapsedSeconds() << "\n\n";
    checkError(cols < 64 || rows < 64, "Image dimensions too low");  // main.cpp:161

    const Watermark engine(rows, cols, inir.Get("paths", "watermark", ""), p, psnr, device);
    MaskRun runs[2] = {{MASK_TYPE::NVF, "NVF", "_W_NVF", {}, {}, {}}, {MASK_TYPE::ME, "ME", "_W_ME", {}, {}, {}}};
    const auto parameters = [&] { cout << " columns and parameters:\np = " << p << "  PSNR(dB) = " << psnr << "\n"; };

    for (MaskRun& r : runs) engine.makeWatermark(image, rgbImage, r.strength, r.type);  // warm-up (main.cpp:169-170)
    for (MaskRun& r : runs) {
        const double secs = averageSeconds(loops, [&] { r.marked = engine.makeWatermark(image, rgbImage, r.strength, r.type); });
        cout << "Watermark strength (parameter a): " << r.strength << "\nCalculation of " << r.name << " mask with " << rows << " rows and " << cols;
        parameters();
        cout << executionTime(showFps, secs) << "\n\n";
    }
    for (MaskRun& r : runs) {
        r.host.resize(3 * n);
        r.marked.host(r.host.data());  // the unquantised watermarked image
        r.markedGray = wm::Image::fromHost(rgb2gray(r.host, n).data(), rows, cols, 1, device);
        engine.detectWatermark(r.markedGray, r.type);  // warm-up
    }
    for (MaskRun& r : runs) {
        const double secs = averageSeconds(loops, [&] { r.correlation = engine.detectWatermark(r.markedGray, r.type); });
        cout << "Calculation of the watermark correlation (" << r.name << ") of an image with " << rows << " rows and " << cols;
        parameters();
        cout << executionTime(showFps, secs) << "\n\n";
    }
    for (const MaskRun& r : runs) {
        char line[64];
        std::snprintf(line, sizeof line, "Correlation [%s]: %.16f\n", r.name, r.correlation);
        cout << line;
    }

    if (inir.GetBoolean("options", "save_watermarked_files_to_disk", false)) {  // main.cpp:229-240 (.as(u8): truncation)
        cout << "\nSaving watermarked files to disk...\n";
        for (const MaskRun& r : runs) {
            std::vector<uint8_t> interleaved(3 * n);
            for (int ch = 0; ch < 3; ++ch)
                for (size_t i = 0; i < n; ++i) interleaved[3 * i + ch] = (uint8_t)r.host[ch * n + i];
            string name = addSuffixBeforeExtension(imageFile, r.suffix);
            const auto dot = name.find_last_of('.');
            name = (dot == string::npos ? name : name.substr(0, dot)) + ".ppm";
            write_ppm(name, (int)rows, (int)cols, interleaved.data());
        }
        cout << "Successully saved to disk\n";
    }
    return EXIT_SUCCESS;
}

// ---- raw yuv420p streams ---------------------------------------------------------------------------------
struct YuvReader {
    FILE* f = nullptr;
    int width = 0, height = 0;
    bool y4m = false;
    string header;
    bool open(const string& path, int w, int h)
    {
        f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        if (path.size() > 4 && path.substr(path.size() - 4) == ".y4m") {
            y4m = true;
            char line[512];
            if (!std::fgets(line, sizeof line, f)) return false;
            header = line;
            if (header.rfind("YUV4MPEG2", 0) != 0) return false;
            for (size_t i = 0; i < header.size(); ++i) {
                if (header[i] == ' ' && i + 1 < header.size()) {
                    if (header[i + 1] == 'W') width = std::atoi(&header[i + 2]);
                    if (header[i + 1] == 'H') height = std::atoi(&header[i + 2]);
                    if (header[i + 1] == 'C' && header.compare(i + 2, 3, "420") != 0) return false;  // only 4:2:0 (main.cpp:458-459)
                }
            }
        } else { width = w; height = h; }
        return width > 0 && height > 0;
    }
    bool next(uint8_t* y, uint8_t* uv)
    {
        if (y4m) {
            char line[128];
            if (!std::fgets(line, sizeof line, f)) return false;  // "FRAME\n"
        }
        const size_t ny = (size_t)width * height, nuv = 2 * ((size_t)(width / 2) * (height / 2));
        if (std::fread(y, 1, ny, f) != ny) return false;
        return std::fread(uv, 1, nuv, f) == nuv;
    }
    ~YuvReader() { if (f) std::fclose(f); }
};

static int testForVideo(const string& videoFile, const INIReader& inir, const int p, const float psnr, const int device)
{
    const bool showFps = inir.GetBoolean("options", "execution_time_in_fps", false);
    const int watermarkInterval = (int)inir.GetInteger("parameters_video", "watermark_interval", 30);
    YuvReader in;
    checkError(!in.open(videoFile, (int)inir.GetInteger("parameters_video", "video_width", 0), (int)inir.GetInteger("parameters_video", "video_height", 0)),
               "ERROR: Failed to open input video file (need a yuv420p .y4m, or .yuv with video_width/video_height)");
    const int width = in.width, height = in.height;
    const size_t ny = (size_t)width * height, nuv = 2 * ((size_t)(width / 2) * (height / 2));
    const Watermark watermarkObj(height, width, inir.Get("paths", "watermark", ""), p, psnr, device);
    wm_ctx* ctx = watermarkObj.handle();
    constexpr int SLOTS = 3;  // frames in flight
    checkError(wm_configure(ctx, SLOTS, 1) != WM_OK, "ERROR: wm_configure failed");

    struct InFlight { uint8_t* y; uint8_t* uv; float a; float corr; int status; int frame; bool busy; bool marked; };
    std::vector<InFlight> ring(SLOTS);
    for (auto& s : ring) {
        s.y = (uint8_t*)wm_host_alloc(ny);     // pinned (the reference's CL_MEM_ALLOC_HOST_PTR buffer, main.cpp:273-275)
        s.uv = (uint8_t*)wm_host_alloc(nuv);
        s.busy = false;
        checkError(!s.y || !s.uv, "ERROR: pinned allocation failed");
    }
    auto plane_of = [&](uint8_t* y) {
        wm_plane pl{};
        pl.data = y; pl.rows = height; pl.cols = width; pl.channels = 1; pl.dtype = WM_U8; pl.mem = WM_MEM_HOST; pl.frames = 1; pl.pitch = width;
        return pl;
    };

    const string outPath = inir.Get("parameters_video", "encode_watermark_file_path", "");
    const bool doEmbed = outPath != "";
    const bool doDetect = !doEmbed && inir.GetBoolean("parameters_video", "watermark_detection", false);
    if (!doEmbed && !doDetect) return EXIT_SUCCESS;
    FILE* out = nullptr;
    bool outY4m = false;
    if (doEmbed) {
        out = outPath == "-" ? stdout : std::fopen(outPath.c_str(), "wb");
        checkError(!out, "Error: Could not open output");
        outY4m = outPath.size() > 4 && outPath.substr(outPath.size() - 4) == ".y4m";
        if (outY4m) std::fputs(in.y4m ? in.header.c_str() : ("YUV4MPEG2 W" + std::to_string(width) + " H" + std::to_string(height) + " F30:1 Ip C420\n").c_str(), out);
    }
    auto retire = [&](InFlight& s, int slot) {  // in frame order: slots are reused round-robin
        if (!s.busy) return;
        if (s.marked) checkError(wm_sync(ctx, slot) < 0, string("ERROR: ") + wm_last_error(ctx));
        if (doEmbed) {
            if (outY4m) std::fputs("FRAME\n", out);
            std::fwrite(s.y, 1, ny, out);   // Y (watermarked or untouched), then U and V as-is (main.cpp:359-386)
            std::fwrite(s.uv, 1, nuv, out);
        } else if (s.marked) {
            cout << "Correlation for frame: " << s.frame << ": " << s.corr << "\n";  // main.cpp:407
        }
        s.busy = false;
    };

    timer::start();
    int framesCount = 0;
    for (;;) {
        const int slot = framesCount % SLOTS;
        InFlight& s = ring[slot];
        retire(s, slot);
        if (!in.next(s.y, s.uv)) break;
        s.frame = framesCount; s.busy = true;
        s.marked = framesCount % watermarkInterval == 0;  // main.cpp:346,395
        if (s.marked) {
            const wm_plane pl = plane_of(s.y);
            int rc;
            if (doEmbed) rc = wm_embed(ctx, WM_MASK_ME, &pl, &pl, &pl, &s.a, &s.status, slot);  // makeWatermark(frame, frame, ME)
            else rc = wm_detect(ctx, WM_MASK_ME, &pl, &s.corr, &s.status, slot);
            checkError(rc < 0, string("ERROR: ") + wm_last_error(ctx));
        }
        framesCount++;
    }
    for (int k = 0; k < SLOTS; ++k) { const int slot = (framesCount + k) % SLOTS; retire(ring[slot], slot); }
    timer::end();
    if (out && out != stdout) std::fclose(out);
    for (auto& s : ring) { wm_host_free(s.y); wm_host_free(s.uv); }
    if (doEmbed) std::cerr << "\nWatermark embedding total execution time: " << executionTime(false, timer::elapsedSeconds()) << "\n";
    else {
        cout << "\nWatermark detection total execution time: " << executionTime(false, timer::elapsedSeconds()) << "\n";
        cout << "\nWatermark detection average execution time per frame: " << executionTime(showFps, timer::elapsedSeconds() / (framesCount ? framesCount : 1)) << "\n";
    }
    return EXIT_SUCCESS;
}

int main(int argc, char** argv)
{
    const INIReader inir(argc > 1 ? argv[1] : "settings.ini");
    checkError(inir.ParseError() < 0, "Could not load settings.ini file");
    int device = (int)inir.GetInteger("options", "opencl_device", 0);
    if (device < 0 || device >= wm_device_count()) {
        cout << "NOTE: Invalid OpenCL device specified, using default 0" << "\n";  // main.cpp:76
        device = 0;
    }
    cout << wm_version() << ", device " << device << "\n\n";
    const int p = (int)inir.GetInteger("parameters", "p", -1);
    const float psnr = inir.GetFloat("parameters", "psnr", -1.0f);
    checkError(p != 3, "For now, only p=3 is allowed");     // main.cpp:89
    checkError(psnr <= 0, "PSNR must be a positive number");  // main.cpp:96
    try {
        const string videoFile = inir.Get("paths", "video", "");
        const int code = videoFile != "" ? testForVideo(videoFile, inir, p, psnr, device) : testForImage(inir, p, psnr, device);
        exitProgram(code);
    } catch (const std::exception& ex) {
        cout << ex.what() << "\n";
        exitProgram(EXIT_FAILURE);
    }
    exitProgram(EXIT_SUCCESS);
}
