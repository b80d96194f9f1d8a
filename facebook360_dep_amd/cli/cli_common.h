// Host-side plumbing shared by the DerpCLI / TemporalBilateralFilter / UpsampleDisparity
// executables: gflags-style flag parsing, glog-style logging with FATAL = exit(1), a small JSON
// reader for rig files, PNG (zlib) and PFM I/O, directory helpers. Plain C++17 + zlib; the
// reference uses gflags, glog, folly, boost::filesystem and OpenCV imgcodecs for the same jobs.
// All computation goes through the C-ABI in include/derp_hip.h.
#pragma once
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/derp_hip.h"
#include "image_codecs.h"

namespace cli {
namespace fs = std::filesystem;

// ---------------------------------------------------------------- logging (glog look-alike)
// --log_dir=<dir>: every line also goes to <dir>/<program>.INFO, the name glog's symlink has and the
// reference's tests read back (scripts/test/test_derp_cli.py:50-62)
inline FILE*& log_file() {
  static FILE* f = nullptr;
  return f;
}
inline void vlog(char sev, const char* file, int line, const std::string& msg) {
  time_t t = time(nullptr);
  struct tm tmv;
  localtime_r(&t, &tmv);
  const char* base = strrchr(file, '/');
  char head[256];
  snprintf(head, sizeof head, "%c%02d%02d %02d:%02d:%02d %s:%d] ", sev, tmv.tm_mon + 1, tmv.tm_mday, tmv.tm_hour,
           tmv.tm_min, tmv.tm_sec, base ? base + 1 : file, line);
  fprintf(stderr, "%s%s\n", head, msg.c_str());
  if (log_file()) {
    fprintf(log_file(), "%s%s\n", head, msg.c_str());
    fflush(log_file());
  }
}
#define LOG_INFO(msg) cli::vlog('I', __FILE__, __LINE__, (msg))
#define LOG_WARNING(msg) cli::vlog('W', __FILE__, __LINE__, (msg))
// a fatal error raised on an I/O worker thread travels to the thread that waits for the batch (IoBatch::wait)
// instead of ending the process from the worker while the main thread may be inside a HIP call
struct WorkerFatal : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline bool& on_worker_thread() {
  static thread_local bool v = false;
  return v;
}
// LOG(FATAL): the reference aborts; callers only look at "exit status != 0" (system_util.py:302-345)
#define LOG_FATAL(msg)                                \
  do {                                                \
    if (cli::on_worker_thread()) {                    \
      throw cli::WorkerFatal(msg);                    \
    }                                                 \
    cli::vlog('F', __FILE__, __LINE__, (msg));        \
    exit(1);                                          \
  } while (0)
#define CHECK_MSG(cond, msg)                                             \
  do {                                                                   \
    if (!(cond)) {                                                       \
      LOG_FATAL(std::string("Check failed: " #cond " ") + (msg));        \
    }                                                                    \
  } while (0)
#define DERP_OK(ctx, expr)                                                            \
  do {                                                                                \
    if ((expr) != 0) {                                                                \
      LOG_FATAL(std::string(#expr " failed: ") + derp_last_error(ctx));               \
    }                                                                                 \
  } while (0)

inline std::string fmt(const char* f, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

// ---------------------------------------------------------------- flags (gflags look-alike)
struct Flags {
  struct Def {
    std::string type, value, help;
  };
  std::map<std::string, Def> defs;
  std::vector<std::string> order;

  void def(const std::string& type, const std::string& name, const std::string& dflt, const std::string& help) {
    defs[name] = {type, dflt, help};
    order.push_back(name);
  }
  void str(const std::string& n, const std::string& d, const std::string& h) { def("string", n, d, h); }
  void i32(const std::string& n, int d, const std::string& h) { def("int32", n, std::to_string(d), h); }
  void dbl(const std::string& n, double d, const std::string& h) {
    std::ostringstream ss;
    ss.precision(17);
    ss << d;
    def("double", n, ss.str(), h);
  }
  void boolean(const std::string& n, bool d, const std::string& h) { def("bool", n, d ? "true" : "false", h); }

  std::string s(const std::string& n) const { return defs.at(n).value; }
  static int int_base(const std::string& v) { return v.size() > 1 && v[0] == '0' && (v[1] == 'x' || v[1] == 'X') ? 16 : 10; }
  int i(const std::string& n) const { return (int)strtoll(defs.at(n).value.c_str(), nullptr, int_base(defs.at(n).value)); }
  double d(const std::string& n) const { return atof(defs.at(n).value.c_str()); }
  bool b(const std::string& n) const {
    const std::string& v = defs.at(n).value;
    return v == "true" || v == "1" || v == "t" || v == "yes" || v == "y";
  }
  void set(const std::string& n, const std::string& v) { defs.at(n).value = v; }

  void set_checked(const std::string& name, const std::string& value, bool from_file) {
    auto it = defs.find(name);
    if (it == defs.end()) {
      if (from_file) {
        return;  // gflags --undefok-like leniency for shared flagfiles
      }
      fprintf(stderr, "ERROR: unknown command line flag '%s'\n", name.c_str());
      exit(1);
    }
    if (it->second.type == "int32" || it->second.type == "double") {
      char* end = nullptr;
      if (it->second.type == "int32") {  // gflags: strtoll, base 16 after "0x" and 10 otherwise, the whole token, 32-bit range
        errno = 0;
        const long long v = strtoll(value.c_str(), &end, int_base(value));
        if (errno == ERANGE || v < INT32_MIN || v > INT32_MAX) {
          end = nullptr;
        }
      } else {
        strtod(value.c_str(), &end);
      }
      if (value.empty() || !end || *end) {
        fprintf(stderr, "ERROR: illegal value '%s' specified for %s flag '%s'\n", value.c_str(),
                it->second.type.c_str(), name.c_str());
        exit(1);
      }
    }
    it->second.value = value;
  }

  void parse_tokens(const std::vector<std::string>& toks, bool from_file) {
    for (size_t k = 0; k < toks.size(); ++k) {
      std::string a = toks[k];
      if (a.rfind("--", 0) == 0) {
        a = a.substr(2);
      } else if (a.rfind("-", 0) == 0) {
        a = a.substr(1);
      } else {
        continue;
      }
      std::string name = a, value;
      bool has_value = false;
      const size_t eq = a.find('=');
      if (eq != std::string::npos) {
        name = a.substr(0, eq);
        value = a.substr(eq + 1);
        has_value = true;
      }
      if (name == "flagfile") {
        if (!has_value && k + 1 < toks.size()) {
          value = toks[++k];
        }
        parse_file(value);
        continue;
      }
      if (name == "help" || name == "helpshort") {
        usage();
        exit(0);
      }
      if (name == "helpxml") {  // gflags' machine-readable flag table (name / meaning / default / type)
        helpxml();
        exit(0);
      }
      auto it = defs.find(name);
      if (it == defs.end() && name.rfind("no", 0) == 0 && defs.count(name.substr(2)) &&
          defs[name.substr(2)].type == "bool") {
        defs[name.substr(2)].value = "false";
        continue;
      }
      if (it != defs.end() && it->second.type == "bool" && !has_value) {
        it->second.value = "true";
        continue;
      }
      if (!has_value) {
        if (k + 1 < toks.size()) {
          value = toks[++k];
        } else if (from_file) {
          continue;  // test flagfiles list bare I/O flag names (res/test/derp_cli.flags)
        } else {
          fprintf(stderr, "ERROR: flag '--%s' is missing its argument\n", name.c_str());
          exit(1);
        }
      }
      set_checked(name, value, from_file);
    }
  }
  void parse_file(const std::string& path) {
    std::ifstream f(path);
    if (!f.good()) {
      fprintf(stderr, "ERROR: can't open flagfile %s\n", path.c_str());
      exit(1);
    }
    std::vector<std::string> toks;
    std::string line;
    while (std::getline(f, line)) {
      const size_t a = line.find_first_not_of(" \t\r");
      if (a == std::string::npos || line[a] == '#') {
        continue;
      }
      const size_t b = line.find_last_not_of(" \t\r");
      // one flag per line: keep "--name=value" / "--name" whole
      parse_tokens({line.substr(a, b - a + 1)}, true);
    }
  }
  std::string usage_msg;
  void usage() const {
    printf("%s\n", usage_msg.c_str());
    for (const auto& n : order) {
      const Def& d = defs.at(n);
      printf("    -%s (%s) type: %s default: %s\n", n.c_str(), d.help.c_str(), d.type.c_str(), d.value.c_str());
    }
  }
  static std::string xml_escape(const std::string& v) {
    std::string o;
    for (char ch : v) {
      o += ch == '&' ? "&amp;" : ch == '<' ? "&lt;" : ch == '>' ? "&gt;" : std::string(1, ch);
    }
    return o;
  }
  std::map<std::string, std::string> declared;  // defaults as declared, before any command-line value
  std::string program;
  void helpxml() const {
    printf("<?xml version=\"1.0\"?>\n<AllFlags>\n<program>%s</program>\n<usage>%s</usage>\n", program.c_str(),
           xml_escape(usage_msg).c_str());
    for (const auto& n : order) {
      const Def& d = defs.at(n);
      const auto it = declared.find(n);
      printf("<flag><file>%s</file><name>%s</name><meaning>%s</meaning><default>%s</default><current>%s</current>"
             "<type>%s</type></flag>\n",
             program.c_str(), n.c_str(), xml_escape(d.help).c_str(),
             xml_escape(it == declared.end() ? d.value : it->second).c_str(), xml_escape(d.value).c_str(), d.type.c_str());
    }
    printf("</AllFlags>\n");
  }
  void parse(int argc, char** argv) {
    program = std::filesystem::path(argv[0]).filename().string();
    for (const auto& n : order) {
      declared[n] = defs.at(n).value;
    }
    // glog flags the pipeline passes (res/flags/*.flags); logging always goes to stderr as well
    str("log_dir", "", "glog: directory for <program>.INFO");
    boolean("alsologtostderr", false, "glog: accepted");
    boolean("logtostderr", false, "glog: accepted");
    i32("stderrthreshold", 2, "glog: accepted");
    i32("v", 0, "glog: accepted");
    i32("minloglevel", 0, "glog: accepted");
    std::vector<std::string> toks(argv + 1, argv + argc);
    parse_tokens(toks, false);
    if (!s("log_dir").empty()) {
      std::error_code ec;
      std::filesystem::create_directories(s("log_dir"), ec);
      const std::string prog = std::filesystem::path(argv[0]).filename().string();
      log_file() = fopen((std::filesystem::path(s("log_dir")) / (prog + ".INFO")).c_str(), "w");
    }
    // SystemUtil.cpp:78-97: log every project flag at start
    for (const auto& n : order) {
      LOG_INFO("--" + n + "=" + defs.at(n).value);
    }
  }
};

// ---------------------------------------------------------------- JSON (rig files only)
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double num = 0;
  bool boolean = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* find(const std::string& k) const {
    for (const auto& kv : obj) {
      if (kv.first == k) {
        return &kv.second;
      }
    }
    return nullptr;
  }
  const Json& at(const std::string& k) const {
    const Json* j = find(k);
    if (!j) {
      LOG_FATAL("rig JSON: missing key '" + k + "'");
    }
    return *j;
  }
};
struct JsonParser {
  const std::string& s;
  size_t p = 0;
  int depth = 0;
  explicit JsonParser(const std::string& text) : s(text) {}
  char peek() const { return p < s.size() ? s[p] : '\0'; }  // never past the text, whatever the file holds
  void ws() {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) {
      ++p;
    }
  }
  [[noreturn]] void bad(const char* what) { LOG_FATAL(fmt("rig JSON parse error at offset %zu: %s", p, what)); }
  Json value() {
    ws();
    if (p >= s.size()) {
      bad("unexpected end");
    }
    if (++depth > 64) {
      bad("nesting deeper than 64 levels");
    }
    struct Leave {
      int& d;
      ~Leave() { --d; }
    } leave{depth};
    Json j;
    const char c = s[p];
    if (c == '{') {
      j.kind = Json::Obj;
      ++p;
      ws();
      if (peek() == '}') {
        ++p;
        return j;
      }
      for (;;) {
        ws();
        Json k = value();
        if (k.kind != Json::Str) {
          bad("object key must be a string");
        }
        ws();
        if (peek() != ':') {
          bad("expected ':'");
        }
        ++p;
        j.obj.emplace_back(k.str, value());
        ws();
        if (peek() == ',') {
          ++p;
          continue;
        }
        if (peek() == '}') {
          ++p;
          break;
        }
        bad("expected ',' or '}'");
      }
    } else if (c == '[') {
      j.kind = Json::Arr;
      ++p;
      ws();
      if (peek() == ']') {
        ++p;
        return j;
      }
      for (;;) {
        j.arr.push_back(value());
        ws();
        if (peek() == ',') {
          ++p;
          continue;
        }
        if (peek() == ']') {
          ++p;
          break;
        }
        bad("expected ',' or ']'");
      }
    } else if (c == '"') {
      j.kind = Json::Str;
      ++p;
      while (p < s.size() && s[p] != '"') {
        if (s[p] == '\\' && p + 1 < s.size()) {
          ++p;
          switch (s[p]) {
            case 'n': j.str += '\n'; break;
            case 't': j.str += '\t'; break;
            case 'u':
              if (p + 4 >= s.size()) {
                bad("truncated \\u escape");
              }
              p += 4;
              j.str += '?';
              break;
            default: j.str += s[p];
          }
        } else {
          j.str += s[p];
        }
        ++p;
      }
      if (p >= s.size()) {
        bad("unterminated string");
      }
      ++p;
    } else if (s.compare(p, 4, "true") == 0) {
      j.kind = Json::Bool;
      j.boolean = true;
      p += 4;
    } else if (s.compare(p, 5, "false") == 0) {
      j.kind = Json::Bool;
      p += 5;
    } else if (s.compare(p, 4, "null") == 0) {
      p += 4;
    } else {
      char* end = nullptr;
      j.kind = Json::Num;
      j.num = strtod(s.c_str() + p, &end);
      if (end == s.c_str() + p) {
        bad("unexpected character");
      }
      p = end - s.c_str();
    }
    return j;
  }
};

inline std::string read_file(const std::string& path) {  // "" when the file cannot be read
  std::string out;
  if (FILE* f = fopen(path.c_str(), "rb")) {
    if (fseek(f, 0, SEEK_END) == 0) {
      const long n = ftell(f);
      if (n > 0 && fseek(f, 0, SEEK_SET) == 0) {
        out.resize((size_t)n);
        out.resize(fread(&out[0], 1, (size_t)n, f));
      }
    }
    fclose(f);
  }
  return out;
}

// Camera::loadRig (Camera.cpp:244-258) -> the C-ABI's camera descriptions
inline std::vector<derp_camera_desc> load_rig(const std::string& path) {
  const std::string text = read_file(path);
  CHECK_MSG(!text.empty(), "could not read JSON file: " + path);
  JsonParser jp(text);
  const Json root = jp.value();
  std::vector<derp_camera_desc> out;
  for (const Json& c : root.at("cameras").arr) {
    derp_camera_desc d;
    memset(&d, 0, sizeof d);
    const Json& ver = c.at("version");
    CHECK_MSG((ver.kind == Json::Num ? ver.num : atof(ver.str.c_str())) >= 1.0, "camera version");
    snprintf(d.id, sizeof d.id, "%s", c.at("id").str.c_str());
    const std::string type = c.at("type").str;
    d.type = type == "FTHETA" ? DERP_FTHETA : type == "RECTILINEAR" ? DERP_RECTILINEAR
        : type == "EQUISOLID" ? DERP_EQUISOLID : type == "ORTHOGRAPHIC" ? DERP_ORTHOGRAPHIC : -1;
    CHECK_MSG(d.type >= 0, "unexpected camera type " + type);
    auto vec = [&](const char* key, double* dst, size_t n) {
      const Json& a = c.at(key);
      CHECK_MSG(a.arr.size() == n, std::string("bad vector ") + key);
      for (size_t i = 0; i < n; ++i) {
        dst[i] = a.arr[i].num;
      }
    };
    vec("origin", d.origin, 3);
    vec("forward", d.forward, 3);
    vec("up", d.up, 3);
    vec("right", d.right, 3);
    vec("resolution", d.resolution, 2);
    vec("focal", d.focal, 2);
    if (c.find("principal")) {
      d.has_principal = 1;
      vec("principal", d.principal, 2);
    }
    if (const Json* dist = c.find("distortion")) {
      CHECK_MSG(dist->arr.size() <= 3, "bad distortion");
      d.has_distortion = 1;
      for (size_t i = 0; i < dist->arr.size(); ++i) {
        d.distortion[i] = dist->arr[i].num;
      }
    }
    if (const Json* fov = c.find("fov")) {
      d.has_fov = 1;
      d.fov = fov->num;
    }
    out.push_back(d);
  }
  return out;
}

// image_util::filterDestinations (ImageUtil.cpp:110-125)
inline std::vector<derp_camera_desc> filter_destinations(const std::vector<derp_camera_desc>& rig,
                                                         const std::string& destinations) {
  if (destinations.empty()) {
    return rig;
  }
  std::vector<derp_camera_desc> out;
  std::stringstream ss(destinations);
  std::string dest;
  while (std::getline(ss, dest, ',')) {
    for (const auto& cam : rig) {
      if (dest == cam.id) {
        out.push_back(cam);
      }
    }
  }
  return out;
}

// ---------------------------------------------------------------- filesystem helpers
inline bool is_hidden(const fs::path& p) {
  const std::string n = p.filename().string();
  return !n.empty() && n[0] == '.';
}
inline std::vector<fs::path> visible_files_sorted(const fs::path& dir) {
  std::vector<fs::path> r;
  if (fs::is_directory(dir)) {
    for (const auto& e : fs::directory_iterator(dir)) {
      if (e.is_regular_file() && !is_hidden(e.path())) {
        r.push_back(e.path());
      }
    }
  }
  std::sort(r.begin(), r.end());
  return r;
}
inline std::string first_extension(const fs::path& dir) {  // FilesystemUtil.h:91-95
  const auto files = visible_files_sorted(dir);
  CHECK_MSG(!files.empty(), "no visible files in " + dir.string());
  return files[0].extension().string();
}
inline std::string zero_pad(int x, int n = 6) {  // image_util::intToStringZeroPad
  char b[32];
  snprintf(b, sizeof b, "%0*d", n, x);
  return b;
}
inline fs::path image_path(const fs::path& dir, const std::string& cam, const std::string& frame,
                           const std::string& ext = "") {  // ImageUtil.h:48-56
  const fs::path camDir = dir / cam;
  return camDir / (frame + (ext.empty() ? first_extension(camDir) : ext));
}
// verifyImagePaths (ImageUtil.cpp:63-95)
inline void verify_image_paths(const fs::path& dir, const std::vector<derp_camera_desc>& rig, const std::string& first,
                               const std::string& last) {
  int a = 0, b = 0;
  try {
    a = std::stoi(first);
    b = std::stoi(last);
  } catch (...) {
    LOG_FATAL("Invalid frame name: " + first + " / " + last);
  }
  CHECK_MSG(a <= b, "first <= last");
  CHECK_MSG(!rig.empty(), "rig.size() > 0");
  const std::string ext = first_extension(dir / rig[0].id);
  for (const auto& cam : rig) {
    for (int f = a; f <= b; ++f) {
      const fs::path p = dir / cam.id / (zero_pad(f) + ext);
      CHECK_MSG(fs::is_regular_file(p), "Missing file: " + p.string());
    }
  }
}

// ---------------------------------------------------------------- PFM (CvUtil.cpp:39-73)
inline void write_pfm(const fs::path& path, const float* m, int w, int h) {
  std::ofstream f(path, std::ios::binary);
  f << "Pf\n" << w << " " << h << "\n-1.0\n";
  f.write(reinterpret_cast<const char*>(m), (size_t)w * h * sizeof(float));
  CHECK_MSG(f.good(), "failed to save image: " + path.string());
}
inline bool pfm_size(const fs::path& path, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  std::string magic;
  std::getline(f, magic);
  if (magic != "Pf") {
    return false;
  }
  f >> w >> h;
  return f.good();
}
inline std::vector<float> read_pfm(const fs::path& path, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  CHECK_MSG(f.good(), "cannot load file: " + path.string());
  std::string magic;
  std::getline(f, magic);
  CHECK_MSG(magic == "Pf", "expected 'Pf' in 1-channel .pfm file header: " + path.string());
  double endian;
  f >> w >> h >> endian;
  CHECK_MSG(endian <= 0.0, "only little endian .pfm files supported: " + path.string());
  f.ignore();
  // the header is text from a file: no allocation before it is known to describe what the file holds
  CHECK_MSG(f.good() && w > 0 && h > 0 && w <= (1 << 20) && h <= (1 << 20), "bad .pfm header: " + path.string());
  const std::streamoff at = f.tellg();
  f.seekg(0, std::ios::end);
  const std::streamoff left = f.tellg() - at;
  f.seekg(at);
  CHECK_MSG(left >= (std::streamoff)((size_t)w * h * sizeof(float)), "truncated .pfm file: " + path.string());
  std::vector<float> m((size_t)w * h);
  f.read(reinterpret_cast<char*>(m.data()), m.size() * sizeof(float));
  return m;
}

// ---------------------------------------------------------------- raster input (image_codecs.h) + PNG output
// cv::imread picks its decoder by the file's signature; so does codecs::decode (PNG, JPEG, TIFF, BMP, PNM)
using Png = codecs::Raster;  // w, h, channels (file order R, G, B [, A]), bitdepth 8 / 16 (32 = float samples in f32), px
inline uint32_t be32(const unsigned char* p) {
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
}
inline std::string read_head(const fs::path& path, size_t n) {
  std::ifstream f(path, std::ios::binary);
  std::string head(n, '\0');
  f.read(&head[0], (std::streamsize)n);
  head.resize((size_t)std::max<std::streamsize>(f.gcount(), 0));
  return head;
}
inline bool raster_size(const fs::path& path, int& w, int& h) {
  // the size sits in the first bytes for PNG / BMP / PNM, behind the metadata segments for JPEG, anywhere for TIFF
  const std::string head = read_head(path, 4096);
  if (codecs::probe_size(reinterpret_cast<const unsigned char*>(head.data()), head.size(), w, h)) {
    return true;
  }
  if (head.size() < 4096) {
    return false;
  }
  const std::string all = read_file(path.string());
  return codecs::probe_size(reinterpret_cast<const unsigned char*>(all.data()), all.size(), w, h);
}
inline Png read_raster(const fs::path& path) {
  const std::string data = read_file(path.string());
  CHECK_MSG(!data.empty(), "failed to load image: " + path.string());
  try {
    return codecs::decode(reinterpret_cast<const unsigned char*>(data.data()), data.size());
  } catch (const codecs::Error& e) {
    LOG_FATAL("failed to load image: " + path.string() + " (" + e.what() + ")");
  }
  return Png();
}
inline void write_png(const fs::path& path, const uint16_t* px, int w, int h, int channels, int bitdepth) {
  const int bpp = channels * bitdepth / 8;
  const size_t stride = (size_t)w * bpp;
  std::vector<unsigned char> raw((stride + 1) * h);
  for (int y = 0; y < h; ++y) {
    unsigned char* line = raw.data() + (stride + 1) * y;
    line[0] = 0;
    for (int i = 0; i < w * channels; ++i) {
      const uint16_t v = px[(size_t)y * w * channels + i];
      if (bitdepth == 16) {
        line[1 + 2 * i] = v >> 8;
        line[2 + 2 * i] = v & 255;
      } else {
        line[1 + i] = (unsigned char)v;
      }
    }
  }
  uLongf clen = compressBound(raw.size());
  std::vector<unsigned char> comp(clen);
  CHECK_MSG(compress2(comp.data(), &clen, raw.data(), raw.size(), 3) == Z_OK, "PNG deflate failed");
  std::ofstream f(path, std::ios::binary);
  auto chunk = [&](const char* tag, const unsigned char* body, uint32_t n) {
    unsigned char len[4] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n};
    f.write(reinterpret_cast<char*>(len), 4);
    f.write(tag, 4);
    if (n) {
      f.write(reinterpret_cast<const char*>(body), n);
    }
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(tag), 4);
    if (n) {
      crc = crc32(crc, body, n);
    }
    unsigned char cb[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    f.write(reinterpret_cast<char*>(cb), 4);
  };
  f.write("\x89PNG\r\n\x1a\n", 8);
  unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                            (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h,
                            (unsigned char)bitdepth, (unsigned char)(channels == 1 ? 0 : channels == 3 ? 2 : 6), 0, 0, 0};
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  CHECK_MSG(f.good(), "failed to save image: " + path.string());
}

// cv_util::loadImage<Vec3w> (CvUtil.h:196-284): IMREAD_UNCHANGED -> 16U (8-bit x257) -> BGR
// into a caller buffer of expectW x expectH x 3 (pinned staging memory of the CLI's I/O pipeline)
// convertTo(CV_8U / CV_16U, scale) of a CV_32F image (CvUtil.h:196-207): saturate_cast of the value rounded to the
// nearest integer, ties to even (cvRound); NaN -> 0 like cvRound's conversion of an invalid value saturated
inline unsigned float_to_uint_sat(float v, float scale, unsigned maxv) {
  const double r = nearbyint((double)v * (double)scale);
  return !(r > 0) ? 0u : r >= (double)maxv ? maxv : (unsigned)r;
}
inline void raster_to_bgr16(const Png& p, const fs::path& path, uint16_t* out) {
  if (p.bitdepth == 32) {  // a float image (one channel): convertTo(CV_16U, 65535 / 1.0), COLOR_GRAY2BGR
    CHECK_MSG(p.channels == 1 && p.f32.size() == (size_t)p.w * p.h, "unsupported float image as colour: " + path.string());
    for (size_t i = 0; i < p.f32.size(); ++i) {
      out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = (uint16_t)float_to_uint_sat(p.f32[i], 65535.0f, 65535u);
    }
    return;
  }
  CHECK_MSG(p.bitdepth == 8 || p.bitdepth == 16, "unsupported bit depth of a colour image: " + path.string());
  const int mul = p.bitdepth == 8 ? 257 : 1;  // convertTo(CV_16U, 65535 / 255)
  const size_t n = (size_t)p.w * p.h;
  for (size_t i = 0; i < n; ++i) {
    if (p.channels == 1) {
      out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = p.px[i] * mul;  // COLOR_GRAY2BGR
    } else {                                                           // COLOR_BGRA2BGR drops the alpha
      out[3 * i + 0] = p.px[p.channels * i + 2] * mul;  // B
      out[3 * i + 1] = p.px[p.channels * i + 1] * mul;  // G
      out[3 * i + 2] = p.px[p.channels * i + 0] * mul;  // R
    }
  }
}
inline void load_color_bgr16_into(const fs::path& path, uint16_t* out, int expectW, int expectH) {
  const std::string data = read_file(path.string());
  CHECK_MSG(!data.empty(), "failed to load image: " + path.string());
  const codecs::Bytes bytes{reinterpret_cast<const unsigned char*>(data.data()), data.size()};
  int w = 0, h = 0;
  bool fast = false;
  try {
    fast = codecs::png_fast_bgr16(bytes, out, expectW, expectH, w, h);
  } catch (const codecs::Error& e) {
    LOG_FATAL("failed to load image: " + path.string() + " (" + e.what() + ")");
  }
  if (fast) {
    CHECK_MSG(w == expectW && h == expectH, "image size mismatch: " + path.string());
    return;
  }
  const Png p = read_raster(path);
  CHECK_MSG(p.w == expectW && p.h == expectH, "image size mismatch: " + path.string());
  raster_to_bgr16(p, path, out);
}
inline std::vector<uint16_t> load_color_bgr16(const fs::path& path, int& w, int& h) {
  const Png p = read_raster(path);
  w = p.w;
  h = p.h;
  std::vector<uint16_t> out((size_t)w * h * 3);
  raster_to_bgr16(p, path, out.data());
  return out;
}
// cv_util::loadImage<bool> (CvUtil.h:226-262): to 8-bit, threshold > 127 -> 1 on every channel, then (3 / 4 channels)
// COLOR_BGR[A]2GRAY of the 0 / 1 values — whose fixed-point weights (B 0.114, G 0.587, R 0.299, rounded) give 1 exactly
// when the GREEN channel passed the threshold (G alone rounds to 1, B + R together to 0)
inline std::vector<uint8_t> load_mask(const fs::path& path, int& w, int& h) {
  const Png p = read_raster(path);
  w = p.w;
  h = p.h;
  std::vector<uint8_t> out((size_t)w * h);
  if (p.bitdepth == 32) {  // a float image (one channel): convertTo(CV_8U, 255 / 1.0), then the threshold
    CHECK_MSG(p.channels == 1 && p.f32.size() == out.size(), "unsupported float image as a mask: " + path.string());
    for (size_t i = 0; i < out.size(); ++i) {
      out[i] = float_to_uint_sat(p.f32[i], 255.0f, 255u) > 127;
    }
    return out;
  }
  CHECK_MSG(p.bitdepth == 8 || p.bitdepth == 16, "unsupported bit depth of a mask: " + path.string());
  const int pick = p.channels >= 3 ? 1 : 0;
  for (size_t i = 0; i < out.size(); ++i) {
    unsigned v = p.px[(size_t)p.channels * i + pick];
    if (p.bitdepth == 16) {
      v = (unsigned)lrintf(v * (255.0f / 65535.0f));  // convertTo(CV_8U, 255/65535): saturate_cast rounds
    }
    out[i] = v > 127;
  }
  return out;
}
// cv_util::loadImage<float>: PFM as is; PNG scaled to [0,1]
// ---------------------------------------------------------------- OpenEXR (input)
// What cv::imread reads back where a directory holds the .exr files --output_formats=exr wrote (they sort before the
// .pfm of the same frame, so getFirstExtension-style lookups pick them): single-part scan-line files with exactly
// one FLOAT channel, compression NONE / ZIPS / ZIP. Anything else is refused by name.
struct ExrInfo {
  int w = 0, h = 0, compression = 0;
  size_t tableOffset = 0;
};
inline bool exr_header(const std::string& data, const fs::path& path, ExrInfo& info, bool fatal) {
  auto bad = [&](const std::string& why) {
    if (fatal) {
      LOG_FATAL("unsupported OpenEXR file (" + why + "): " + path.string());
    }
    return false;
  };
  if (data.size() < 12 || memcmp(data.data(), "\x76\x2f\x31\x01", 4) != 0) {
    return bad("bad magic");
  }
  uint32_t version;
  memcpy(&version, data.data() + 4, 4);
  if ((version & 0xff) != 2 || (version & 0x1a00)) {
    return bad("tiled / multi-part / deep");
  }
  size_t pos = 8;
  bool haveChannels = false, haveWindow = false;
  while (pos < data.size() && data[pos] != 0) {
    const size_t e = data.find('\0', pos);
    const size_t e2 = e == std::string::npos ? e : data.find('\0', e + 1);
    if (e2 == std::string::npos || e2 + 5 > data.size()) {
      return bad("truncated header");
    }
    const std::string name = data.substr(pos, e - pos);
    int32_t size;
    memcpy(&size, data.data() + e2 + 1, 4);
    const size_t val = e2 + 5;
    if (size < 0 || val + (size_t)size > data.size()) {
      return bad("truncated header");
    }
    if (name == "channels") {
      const size_t ce = data.find('\0', val);
      int32_t ptype = -1;
      if (ce != std::string::npos && ce + 5 <= val + size) {
        memcpy(&ptype, data.data() + ce + 1, 4);
      }
      // one channel = name\0 + 16 bytes, then the list's terminating \0
      if (ptype != 2 || ce + 17 != val + (size_t)size - 1) {
        return bad("needs exactly one FLOAT channel");
      }
      haveChannels = true;
    } else if (name == "compression" && size == 1) {
      info.compression = (unsigned char)data[val];
    } else if (name == "dataWindow" && size == 16) {
      int32_t b[4];
      memcpy(b, data.data() + val, 16);
      const int64_t ww = (int64_t)b[2] - b[0] + 1, hh = (int64_t)b[3] - b[1] + 1;
      info.w = ww > 0 && ww <= (1 << 20) ? (int)ww : 0;
      info.h = hh > 0 && hh <= (1 << 20) ? (int)hh : 0;
      haveWindow = b[0] == 0 && b[1] == 0;
    }
    pos = val + size;
  }
  if (!haveChannels || !haveWindow || info.w <= 0 || info.h <= 0) {
    return bad("missing channels / dataWindow");
  }
  if (info.compression != 0 && info.compression != 2 && info.compression != 3) {
    return bad("compression other than NONE / ZIPS / ZIP");
  }
  info.tableOffset = pos + 1;
  return true;
}
inline std::string read_file(const fs::path& path) {
  std::ifstream f(path, std::ios::binary);
  CHECK_MSG(f.good(), "failed to load image: " + path.string());
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
inline bool exr_size(const fs::path& path, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  std::string head(4096, '\0');
  f.read(&head[0], (std::streamsize)head.size());
  head.resize((size_t)f.gcount());
  ExrInfo info;
  if (!exr_header(head, path, info, false)) {
    // a header longer than the first 4 KB (many attributes) is still a valid file: parse all of it before giving up
    if (head.size() < 4096 || !exr_header(read_file(path), path, info, false)) {
      return false;
    }
  }
  w = info.w;
  h = info.h;
  return true;
}
inline std::vector<float> read_exr_f32(const fs::path& path, int& w, int& h) {
  const std::string data = read_file(path);
  ExrInfo info;
  exr_header(data, path, info, true);
  w = info.w;
  h = info.h;
  // sizes come from the file: a data window the file's bytes cannot fill (ZIP shrinks float data by a few per cent, and
  // deflate by 1032 : 1 at the very most) is refused before anything is allocated
  CHECK_MSG(w <= (1 << 20) && h <= (1 << 20) && (size_t)w * h * 4 / 1032 <= data.size(), "corrupt OpenEXR header (data window): " + path.string());
  const int lines = info.compression == 3 ? 16 : 1;
  const int blocks = (h + lines - 1) / lines;
  CHECK_MSG(info.tableOffset + (size_t)blocks * 8 <= data.size(), "truncated OpenEXR file: " + path.string());
  std::vector<float> out((size_t)w * h);
  std::vector<unsigned char> tmp;
  for (int b = 0; b < blocks; ++b) {
    uint64_t off;
    memcpy(&off, data.data() + info.tableOffset + (size_t)b * 8, 8);
    // offsets come from the file: no arithmetic on them before they are known to lie inside it (off + 8 would wrap)
    CHECK_MSG(data.size() >= 8 && off <= data.size() - 8, "truncated OpenEXR file: " + path.string());
    int32_t y, size;
    memcpy(&y, data.data() + off, 4);
    memcpy(&size, data.data() + off + 4, 4);
    CHECK_MSG(y >= 0 && y < h && size >= 0 && (size_t)size <= data.size() - 8 - off, "corrupt OpenEXR chunk: " + path.string());
    const int n = std::min(lines, h - y);
    const size_t raw = (size_t)n * w * 4;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(data.data() + off + 8);
    unsigned char* dst = reinterpret_cast<unsigned char*>(out.data() + (size_t)y * w);
    if (info.compression == 0 || (size_t)size == raw) {
      CHECK_MSG((size_t)size == raw, "corrupt OpenEXR chunk: " + path.string());
      memcpy(dst, src, raw);
      continue;
    }
    tmp.resize(raw);
    uLongf got = (uLongf)raw;
    CHECK_MSG(uncompress(tmp.data(), &got, src, (uLong)size) == Z_OK && got == raw, "OpenEXR inflate failed: " + path.string());
    for (size_t i = 1; i < raw; ++i) {  // undo the predictor ...
      tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
    }
    const unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (raw + 1) / 2;  // ... and the byte de-interleave
    for (size_t i = 0; i < raw; ++i) {
      dst[i] = (i & 1) ? *t2++ : *t1++;
    }
  }
  return out;
}

inline std::vector<float> load_float(const fs::path& path, int& w, int& h) {
  if (path.extension() == ".pfm") {
    return read_pfm(path, w, h);
  }
  if (path.extension() == ".exr") {
    return read_exr_f32(path, w, h);
  }
  const Png p = read_raster(path);
  w = p.w;
  h = p.h;
  if (p.bitdepth == 32) {  // a float TIFF: convertTo(CV_32F) of CV_32F is a copy
    return p.f32;
  }
  std::vector<float> out((size_t)w * h);
  const float scale = 1.0f / (p.bitdepth == 16 ? 65535.0f : 255.0f);
  for (size_t i = 0; i < out.size(); ++i) {
    if (p.channels == 1) {
      out[i] = p.px[i] * scale;
    } else {  // COLOR_BGR[A]2GRAY on the scaled floats (order of the float operations as OpenCV's scalar code; unpinned)
      const float r = p.px[(size_t)p.channels * i] * scale, g = p.px[(size_t)p.channels * i + 1] * scale,
                  b = p.px[(size_t)p.channels * i + 2] * scale;
      out[i] = b * 0.114f + g * 0.587f + r * 0.299f;
    }
  }
  return out;
}
inline bool image_size(const fs::path& path, int& w, int& h) {
  return path.extension() == ".pfm" ? pfm_size(path, w, h) : path.extension() == ".exr" ? exr_size(path, w, h) : raster_size(path, w, h);
}
// ---------------------------------------------------------------- OpenEXR (output)
// What cv::imwrite(".exr", CV_32FC1) leaves behind for --output_formats=exr (PyramidLevel.h:515-516,
// CvUtil.cpp:30-37): a single-part scan-line OpenEXR 2 file with one 32-bit FLOAT channel "Y", ZIP compression
// (blocks of 16 scan lines; OpenEXR's default, which OpenCV does not override), increasing-Y line order. Written
// from the published file layout (openexr.com "OpenEXR File Layout"): magic, version, attribute list, line-offset
// table, chunks; ZIP = byte de-interleave + delta predictor + zlib deflate, raw when deflate does not shrink it.
inline void write_exr_f32(const fs::path& path, const float* m, int w, int h) {
  std::string hdr;
  auto put = [&](const void* p, size_t n) { hdr.append(static_cast<const char*>(p), n); };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_f32 = [&](float v) { put(&v, 4); };
  auto attr = [&](const char* name, const char* type, const std::string& value) {
    hdr.append(name).push_back('\0');
    hdr.append(type).push_back('\0');
    put_i32((int32_t)value.size());
    hdr.append(value);
  };
  const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};  // magic, version 2, no flags
  put(magic, 8);
  {
    std::string ch("Y");
    ch.push_back('\0');
    const int32_t pixelType = 2;  // FLOAT
    ch.append(reinterpret_cast<const char*>(&pixelType), 4);
    ch.append(std::string("\0\0\0\0", 4));  // pLinear + 3 reserved bytes
    const int32_t one = 1;
    ch.append(reinterpret_cast<const char*>(&one), 4);  // xSampling
    ch.append(reinterpret_cast<const char*>(&one), 4);  // ySampling
    ch.push_back('\0');                                 // end of the channel list
    attr("channels", "chlist", ch);
  }
  attr("compression", "compression", std::string(1, (char)3));  // ZIP_COMPRESSION
  const int32_t box[4] = {0, 0, w - 1, h - 1};
  attr("dataWindow", "box2i", std::string(reinterpret_cast<const char*>(box), 16));
  attr("displayWindow", "box2i", std::string(reinterpret_cast<const char*>(box), 16));
  attr("lineOrder", "lineOrder", std::string(1, (char)0));  // INCREASING_Y
  {
    const float one = 1.0f, zero2[2] = {0.0f, 0.0f};
    attr("pixelAspectRatio", "float", std::string(reinterpret_cast<const char*>(&one), 4));
    attr("screenWindowCenter", "v2f", std::string(reinterpret_cast<const char*>(zero2), 8));
    attr("screenWindowWidth", "float", std::string(reinterpret_cast<const char*>(&one), 4));
  }
  hdr.push_back('\0');  // end of the header
  (void)put_f32;
  const int kLines = 16;
  const int blocks = (h + kLines - 1) / kLines;
  std::vector<std::string> chunks(blocks);
  std::vector<unsigned char> tmp, packed;
  for (int b = 0; b < blocks; ++b) {
    const int y0 = b * kLines, lines = std::min(kLines, h - y0);
    const size_t raw = (size_t)lines * w * 4;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(m + (size_t)y0 * w);
    tmp.resize(raw);
    {  // even bytes first, odd bytes second; then each byte becomes its difference from the one before
      unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (raw + 1) / 2;
      for (size_t i = 0; i < raw; ++i) {
        *((i & 1) ? t2++ : t1++) = src[i];
      }
      int p = tmp[0];
      for (size_t i = 1; i < raw; ++i) {
        const int d = (int)tmp[i] - p + (128 + 256);
        p = tmp[i];
        tmp[i] = (unsigned char)d;
      }
    }
    uLongf bound = compressBound((uLong)raw);
    packed.resize(bound);
    CHECK_MSG(compress(packed.data(), &bound, tmp.data(), (uLong)raw) == Z_OK, "deflate failed: " + path.string());
    std::string& c = chunks[b];
    const int32_t y = y0;
    const bool shrunk = bound < raw;
    const int32_t size = (int32_t)(shrunk ? bound : raw);
    c.append(reinterpret_cast<const char*>(&y), 4);
    c.append(reinterpret_cast<const char*>(&size), 4);
    c.append(reinterpret_cast<const char*>(shrunk ? packed.data() : src), (size_t)size);
  }
  std::ofstream f(path, std::ios::binary);
  f.write(hdr.data(), (std::streamsize)hdr.size());
  uint64_t off = hdr.size() + (uint64_t)blocks * 8;
  for (int b = 0; b < blocks; ++b) {
    f.write(reinterpret_cast<const char*>(&off), 8);
    off += chunks[b].size();
  }
  for (int b = 0; b < blocks; ++b) {
    f.write(chunks[b].data(), (std::streamsize)chunks[b].size());
  }
  CHECK_MSG(f.good(), "failed to save image: " + path.string());
}

// cv_util::convertTo<uint16_t>(float disparity): x65535, saturate (NaN -> 0), PyramidLevel.h:517-519
inline void write_disparity_png(const fs::path& path, const float* m, int w, int h) {
  std::vector<uint16_t> px((size_t)w * h);
  for (size_t i = 0; i < px.size(); ++i) {
    const float v = m[i] * 65535.0f;
    px[i] = !(v == v) ? 0 : v <= 0 ? 0 : v >= 65535.f ? 65535 : (uint16_t)lrintf(v);
  }
  write_png(path, px.data(), w, h, 1, 16);
}

// getPyramidLevelSizes (Derp.cpp:72-99): level_N -> size of the first image found under it
inline void pyramid_level_sizes(std::map<int, std::pair<int, int>>& sizes, const fs::path& dir) {
  if (!fs::exists(dir)) {
    return;
  }
  for (const auto& e : fs::directory_iterator(dir)) {
    if (!e.is_directory() || is_hidden(e.path())) {
      continue;
    }
    const std::string name = e.path().filename().string();
    if (name.rfind("level_", 0) != 0) {
      continue;
    }
    std::vector<fs::path> files;
    for (const auto& f : fs::recursive_directory_iterator(e.path())) {
      if (f.is_regular_file() && !is_hidden(f.path()) && f.path().extension() != ".tar") {
        files.push_back(f.path());
      }
    }
    if (files.empty()) {
      continue;
    }
    std::sort(files.begin(), files.end());
    int w, h;
    if (image_size(files[0], w, h)) {
      sizes[std::stoi(name.substr(6))] = {w, h};
    }
  }
}

// I/O worker pool of the executables (--threads: -1 = one per hardware thread, 0 = run inline): PNG / PFM
// decode and encode run here while the GPU computes. The reference spends --threads on the compute itself
// (ThreadPool.h); here the compute is the GPU's.
struct IoPool {
  std::vector<std::thread> workers;
  // three classes of work, served in this order: 0 = staging copies the GPU-feeding thread waits for, 1 = result
  // files (their download buffers are recycled two levels later), 2 = everything else (decoding the inputs, which is
  // queued long in advance: a FIFO would park a level's few small writes behind a second of PNG inflation)
  std::deque<std::function<void()>> queue[3];
  std::mutex mu;
  std::condition_variable cvWork, cvIdle;
  int busy = 0, express = 0;
  bool stop = false;
  // CPUs this process may actually use: the scheduler affinity mask and the cgroup CPU quota (a container that sees
  // 256 hardware threads may be allowed 16 CPUs' worth of time; 64 busy workers then throttle the thread that feeds
  // the GPU — measured on such a box: 24 workers 3.2 s, 64 workers 3.6 s for the same 8-frame job)
  static int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
      n = std::min(n, CPU_COUNT(&set));
    }
    long long quota = -1, period = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
      char q[64];
      if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) {
        quota = atoll(q);
      }
      fclose(f);
    } else {  // cgroup v1
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) {
          quota = -1;
        }
        fclose(g);
      }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(g, "%lld", &period) != 1) {
          period = 100000;
        }
        fclose(g);
      }
    }
    if (quota > 0 && period > 0) {
      n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    }
    return std::max(n, 1);
  }
  explicit IoPool(int threads) {
    // -1 = auto: one and a half workers per usable CPU (file reads / writes block), at most 64
    int n = threads < 0 ? usable_cpus() * 3 / 2 : threads;
    n = std::min(n, 64);
    // Priorities do not pre-empt: with every worker inside a 0.1-s PNG inflation, a staging copy or a result file waits
    // for the first of them to finish (measured: up to 60 ms "waited for a free download plane" per coarse level).
    // Two workers of a large pool therefore never take class-2 jobs.
    express = n >= 8 ? 2 : 0;
    for (int i = 0; i < n; ++i) {
      const bool expressOnly = i < express;
      workers.emplace_back([this, expressOnly] {
        on_worker_thread() = true;
        // below the thread that feeds the GPU: a saturated host otherwise stretches its kernel launches
        (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);
        const int classes = expressOnly ? 2 : 3;
        auto mine = [this, classes] {
          for (int c = 0; c < classes; ++c) {
            if (!queue[c].empty()) {
              return true;
            }
          }
          return false;
        };
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu);
            cvWork.wait(lk, [&] { return stop || mine(); });
            if (!mine()) {
              return;
            }
            for (int c = 0; c < classes; ++c) {
              if (!queue[c].empty()) {
                job = std::move(queue[c].front());
                queue[c].pop_front();
                break;
              }
            }
            ++busy;
          }
          job();
          {
            std::lock_guard<std::mutex> lk(mu);
            --busy;
          }
          cvIdle.notify_all();
        }
      });
    }
  }
  bool pending_locked() const { return !queue[0].empty() || !queue[1].empty() || !queue[2].empty(); }
  void submit(std::function<void()> job, int prio = 2) {
    if (workers.empty()) {
      job();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      queue[std::min(std::max(prio, 0), 2)].push_back(std::move(job));
    }
    if (express) {
      cvWork.notify_all();  // notify_one could pick an express worker for a class-2 job, which would go back to sleep
    } else {
      cvWork.notify_one();
    }
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu);
    cvIdle.wait(lk, [this] { return !pending_locked() && busy == 0; });
  }
  ~IoPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cvWork.notify_all();
    for (auto& t : workers) {
      t.join();
    }
  }
};
// a batch of pool jobs that can be waited for on its own (frame f + 1 decoding while frame f's files are written)
struct IoBatch {
  std::mutex mu;
  std::condition_variable cv;
  int pending = 0;
  std::string error;  // first fatal error of a job; re-raised by wait() on the waiting thread
  void add(IoPool& pool, std::function<void()> job, int prio = 2) {
    {
      std::lock_guard<std::mutex> lk(mu);
      ++pending;
    }
    pool.submit([this, job] {
      std::string err;
      try {
        job();
      } catch (const std::exception& e) {
        err = e.what();
        if (err.empty()) {
          err = "I/O job failed";
        }
      }
      // notify while holding the mutex: the waiter may destroy this batch as soon as it sees pending == 0
      std::lock_guard<std::mutex> lk(mu);
      if (!err.empty() && error.empty()) {
        error = err;
      }
      --pending;
      cv.notify_all();
    }, prio);
  }
  bool done() {
    std::lock_guard<std::mutex> lk(mu);
    return pending == 0;
  }
  // the first error any finished job left, raised on the calling thread without waiting for the rest
  void raise_if_failed() {
    std::string err;
    {
      std::lock_guard<std::mutex> lk(mu);
      err = error;
    }
    if (!err.empty()) {
      LOG_FATAL(err);
    }
  }
  void wait() {
    std::string err;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return pending == 0; });
      err.swap(error);
    }
    if (!err.empty()) {
      LOG_FATAL(err);
    }
  }
};

// rows [0, n) split into contiguous chunks over the pool; returns when all of them ran
inline void parallel_rows(IoPool& pool, int n, const std::function<void(int, int)>& body) {
  const int parts = std::max(1, std::min(n, (int)pool.workers.size()));
  if (parts <= 1) {
    body(0, n);
    return;
  }
  IoBatch batch;
  for (int p = 0; p < parts; ++p) {
    const int a = (int)((long long)n * p / parts), b = (int)((long long)n * (p + 1) / parts);
    batch.add(pool, [=, &body] { body(a, b); });
  }
  batch.wait();
}

struct Timer {
  timespec t0;
  Timer() { clock_gettime(CLOCK_MONOTONIC, &t0); }
  double s() const {
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  }
};

}  // namespace cli
