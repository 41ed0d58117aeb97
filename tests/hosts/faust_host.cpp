// Test host for the generated mydsp adapter: plays the part of FaustJuceProcessor (src/FaustJuceProcessor.cpp:
// 319-323 collects the UI zones, :431-437 init, :462-482 pushes parameter values into the zones and calls compute in place).
// The base types below state the interface of the reference's src/faust_support_min.h (Meta / UI / dsp); they are this
// repo's own declarations of that interface.
//   faust_host <in.f32> <out.f32> <frames> <block> <sr> [label=value ...]     (planar stereo float32 files)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#define FAUSTFLOAT float
struct Meta { virtual ~Meta() = default; virtual void declare(const char* key, const char* value) = 0; };
struct UI {
  virtual ~UI() = default;
  virtual void openTabBox(const char* label) = 0;
  virtual void openHorizontalBox(const char* label) = 0;
  virtual void openVerticalBox(const char* label) = 0;
  virtual void closeBox() = 0;
  virtual void addButton(const char* label, FAUSTFLOAT* zone) = 0;
  virtual void addCheckButton(const char* label, FAUSTFLOAT* zone) = 0;
  virtual void addVerticalSlider(const char* label, FAUSTFLOAT* zone, FAUSTFLOAT init, FAUSTFLOAT min, FAUSTFLOAT max, FAUSTFLOAT step) = 0;
  virtual void addHorizontalSlider(const char* label, FAUSTFLOAT* zone, FAUSTFLOAT init, FAUSTFLOAT min, FAUSTFLOAT max, FAUSTFLOAT step) = 0;
  virtual void addNumEntry(const char* label, FAUSTFLOAT* zone, FAUSTFLOAT init, FAUSTFLOAT min, FAUSTFLOAT max, FAUSTFLOAT step) = 0;
  virtual void addHorizontalBargraph(const char* label, FAUSTFLOAT* zone, FAUSTFLOAT min, FAUSTFLOAT max) = 0;
  virtual void addVerticalBargraph(const char* label, FAUSTFLOAT* zone, FAUSTFLOAT min, FAUSTFLOAT max) = 0;
  virtual void declare(FAUSTFLOAT* zone, const char* key, const char* value) = 0;
};
class dsp {
 public:
  virtual ~dsp() = default;
  virtual int getNumInputs() = 0;
  virtual int getNumOutputs() = 0;
  virtual void buildUserInterface(UI* ui) = 0;
  virtual void metadata(Meta* m) = 0;
  virtual void init(int sample_rate) = 0;
  virtual void compute(int count, FAUSTFLOAT** inputs, FAUSTFLOAT** outputs) = 0;
};

#include ZAB_MYDSP_HEADER

struct Zones : UI {
  std::vector<std::pair<std::string, FAUSTFLOAT*>> params;
  int declared = 0;
  void openTabBox(const char*) override {}
  void openHorizontalBox(const char*) override {}
  void openVerticalBox(const char*) override {}
  void closeBox() override {}
  void addButton(const char* l, FAUSTFLOAT* z) override { params.push_back({l, z}); }
  void addCheckButton(const char* l, FAUSTFLOAT* z) override { params.push_back({l, z}); }
  void addVerticalSlider(const char* l, FAUSTFLOAT* z, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT) override { params.push_back({l, z}); }
  void addHorizontalSlider(const char* l, FAUSTFLOAT* z, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT) override { params.push_back({l, z}); }
  void addNumEntry(const char* l, FAUSTFLOAT* z, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT, FAUSTFLOAT) override { params.push_back({l, z}); }
  void addHorizontalBargraph(const char*, FAUSTFLOAT*, FAUSTFLOAT, FAUSTFLOAT) override {}
  void addVerticalBargraph(const char*, FAUSTFLOAT*, FAUSTFLOAT, FAUSTFLOAT) override {}
  void declare(FAUSTFLOAT*, const char*, const char*) override { ++declared; }
};

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
  const long frames = atol(argv[3]), block = atol(argv[4]);
  const int sr = atoi(argv[5]);
  mydsp d;
  Zones ui;
  d.buildUserInterface(&ui);
  if (argc == 6 && frames == 0) {                 // list mode: print the parameter labels
    for (auto& p : ui.params) printf("%s=%g\n", p.first.c_str(), *p.second);
    return 0;
  }
  const int nch = d.getNumInputs();
  std::vector<float> buf((size_t)nch * frames);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) { fprintf(stderr, "cannot read input\n"); return 2; }
  fclose(f);
  d.init(sr);                                     // prepareToPlay
  for (long pos = 0; pos < frames; pos += block) {
    for (int k = 6; k < argc; ++k) {              // processBlock: parameter values -> zones, every block
      const char* eq = strrchr(argv[k], '=');
      if (!eq) continue;
      const std::string label(argv[k], eq - argv[k]);
      bool hit = false;
      for (auto& p : ui.params) if (p.first == label) { *p.second = (float)atof(eq + 1); hit = true; }
      if (!hit) { fprintf(stderr, "no parameter '%s'\n", label.c_str()); return 2; }
    }
    const int n = (int)((frames - pos) < block ? (frames - pos) : block);
    std::vector<float*> ch(nch);
    for (int c = 0; c < nch; ++c) ch[c] = buf.data() + (size_t)c * frames + pos;
    d.compute(n, ch.data(), ch.data());           // in place, as the JUCE buffer is
  }
  f = fopen(argv[2], "wb");
  fwrite(buf.data(), sizeof(float), buf.size(), f);
  fclose(f);
  return 0;
}
