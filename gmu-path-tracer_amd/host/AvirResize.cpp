// Square RGBA8 resize exactly as the reference's call  avir::CImageResizer<avir::fpclass_float8_dil>(8).resizeImage(src, n, n, 0, dst, m, m, 4, 0)
// (Source/Scene.cpp:269-279; avir 2.4, Include/avir/avir.h, avir_dil.h, avir_float8_avx.h in the reference tree).
//
// A restatement of avir's pipeline for THIS call only -- 8-bit in, 8-bit out, four channels, square to square, automatic step (k = 0),
// default parameters, one thread -- written for one channel plane at a time with plain arrays; it is not avir's class structure.  What has to be
// avir's, because every output byte depends on it:
//   * the filter design in binary64 (peaked-cosine windowed sinc low-pass filters, the 65-bin equaliser of the correction filter, the
//     fractional-delay filter bank), incl. its recursive sine oscillators and its normalise / trim / normalise order;
//   * the choice among the four build modes (filter + interpolator combined or not, 0th or 1st order interpolation) by avir's integer
//     complexity model, for the horizontal pass and again for the vertical pass (which sees the filters the horizontal pass created);
//   * the step geometry: edge pixels, prefix / suffix lengths, edge replication before every step;
//   * binary32 arithmetic in the order of the float8 (AVX) code: eight partial sums over the taps (tap i goes to sum i mod 8),
//     combined as ((s0+s4)+(s1+s5))+((s2+s6)+(s3+s7)); round-half-even and clamp at the end.
// The binary64 design uses libm's sin / cos / pow / exp / sqrt: results are bit-identical to the reference's on the same libm (the build
// container and the GPU boxes share one image; tests/golden/texture_ref.npz pins it).
#include "TextureLoader.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

namespace gmupt {
namespace {

constexpr double kPi = 3.1415926535897932;     // avir.h:76 AVIR_PI
constexpr double kPiD2 = 1.5707963267948966;   // avir.h:83
constexpr int kAlign = 8;                      // float8: filters are padded to a multiple of eight taps (fpclass_float8_dil::elalign)
constexpr int kChannels = 4;

// default parameters (avir.h: CImageResizerParamsDef, CImageResizerParams)
constexpr double kCorrFltAlpha = 1.0, kCorrFltLen = 6.30770, kIntFltAlpha = 2.27825, kIntFltCutoff = 0.75493, kIntFltLen = 18.0;
constexpr double kLPFltAlpha = 3.40127, kLPFltBaseLen = 7.78, kLPFltCutoffMult = 0.78797;
constexpr double kHBFltAlpha = 1.75395, kHBFltCutoff = 0.40356, kHBFltLen = 22.0;
constexpr int kEdgePixels = 3;                 // CImageResizerFilterStep::EdgePixelCountDef

// ---- sine oscillator and window (avir.h: CSineGen, CDSPWindowGenPeakedCosine)
struct SineGen {
    double v1, v2, incr;
    SineGen(double si, double ph) : v1(std::sin(ph)), v2(std::sin(ph - si)), incr(2.0 * std::cos(si)) {}
    double next() { const double r = v1; v1 = incr * r - v2; v2 = r; return r; }
};
struct PeakedCosineWindow {
    double alpha, len2; int n; SineGen w;
    PeakedCosineWindow(double a, double l2) : alpha(a), len2(l2), n(0), w(kPiD2 / l2, kPi * 0.5) {}
    double next() { const double h = std::pow(n / len2, alpha); n++; return w.next() * (1.0 - h); }
};

// ---- windowed-sinc low-pass filter (avir.h: CDSPPeakedCosineLPF)
struct LowPass {
    int fl2, len; double len2, freq, alpha;
    LowPass(double l2, double f, double a) : fl2((int)std::ceil(l2) - 1), len(2 * ((int)std::ceil(l2) - 1) + 1), len2(l2), freq(f), alpha(a) {}
    void generate(double* out, double dcGain) const
    {
        PeakedCosineWindow wf(alpha, len2);
        SineGen f2(freq, 0.0);
        double* op = out + fl2; double* op2 = op;
        f2.next();
        *op = freq * wf.next() / kPi;
        double s = *op;
        for (int t = 1; t <= fl2; t++) {
            const double v = f2.next() * wf.next() / t / kPi;
            op++; op2--;
            *op = v; *op2 = v;
            s += *op + *op2;
        }
        s = dcGain / s;
        for (int t = 0; t < len; t++) op2[t] = op2[t] * s;
    }
};

// a designed filter in binary64 with the parameters avir compares it by (CFltBuffer)
struct DesignedFilter { std::vector<double> taps; double len2 = 0.0, freq = 0.0, alpha = 0.0, dcGain = 0.0;
                        bool sameDesign(const DesignedFilter& o) const { return len2 == o.len2 && freq == o.freq && alpha == o.alpha && dcGain == o.dcGain; } };

void normalize(double* p, int l, double dcGain) { double s = 0.0; for (int i = 0; i < l; i++) s += p[i]; s = dcGain / s; for (int i = 0; i < l; i++) p[i] = p[i] * s; }

// leading / trailing taps below 1e-5 are cut off symmetrically (avir.h: optimizeFIRFilter)
void trimFilter(std::vector<double>& f, int& latency)
{
    for (int i = 0; i <= latency; i++)
        if (std::fabs(f[(size_t)i]) >= 0.00001 || i == latency) {
            if (i > 0) { const int n = (int)f.size() - 2 * i; for (int k = 0; k < n; k++) f[(size_t)k] = f[(size_t)(k + i)]; f.resize((size_t)n); latency -= i; }
            break;
        }
}

// frequency response of a binary32 filter at theta (avir.h: calcFIRFilterResponse, latency 0)
void response(const float* flt, int len, double th, double& re0, double& im0)
{
    const double sincr = 2.0 * std::cos(th);
    double c1 = 1.0, s1 = 0.0, c2 = std::cos(-th), s2 = std::sin(-th), re = 0.0, im = 0.0;
    for (int i = 0; i < len; i++) {
        re += c1 * flt[i]; im += s1 * flt[i];
        double t = c1; c1 = sincr * c1 - c2; c2 = t;
        t = s1; s1 = sincr * s1 - s2; s2 = t;
    }
    re0 = re; im0 = im;
}

// ---- the equaliser that designs the correction filter (avir.h: CDSPFIREQ with linear bands from 0 to MaxFreq)
struct Equalizer {
    int z, zi, z2, bands; std::vector<double> center, k1, k2; bool lastVirtual;
    void init(double sampleRate, double filterLength, int bandCount, double maxFreq, double wfAlpha)
    {
        bands = bandCount;
        z = (int)std::ceil(filterLength * 0.5); zi = z + (z & 1); z2 = z * 2;
        center.assign((size_t)bands, 0.0);
        std::vector<double> osc((size_t)z2), win((size_t)z);
        for (int i = 0; i < z; i++) { osc[(size_t)(2 * i)] = 0.0; osc[(size_t)(2 * i + 1)] = 1.0; }
        { PeakedCosineWindow wf(wfAlpha, filterLength * 0.5); for (int i = 1; i <= z; i++) win[(size_t)(z - i)] = wf.next(); }
        k1.assign((size_t)(zi * bands), 0.0); k2.assign((size_t)(zi * bands), 0.0);     // MinFreq = 0: no first virtual band; room for a last one is not needed (bands - 1 kernels + at most 1)
        const double mo = (maxFreq - 0.0) / (bands - 1);
        double f = 0.0, x1 = 0.0;
        center[0] = 0.0; f = f * 1.0 + mo;
        double* kb1 = k1.data(); double* kb2 = k2.data();
        for (int i = 1; i < bands; i++) {
            const double x2 = f * 2.0 / sampleRate;
            center[(size_t)i] = x2;
            bandKernel(x1, x2, kb1, kb2, osc.data(), win.data());
            kb1 += zi; kb2 += zi; x1 = x2; f = f * 1.0 + mo;
        }
        if (x1 < 1.0) { lastVirtual = true; bandKernel(x1, 1.0, kb1, kb2, osc.data(), win.data()); } else lastVirtual = false;
    }
    void bandKernel(double x1, double x2, double* kb1, double* kb2, double* osc, const double* win) const
    {
        const double incr = kPi * x2, coeff = 2.0 * std::cos(incr);
        double s2v = std::sin(incr * (-z + 1)), c2v = std::sin(incr * (-z + 1) + kPi * 0.5);
        osc[0] = std::sin(incr * -z); osc[1] = std::sin(incr * -z + kPi * 0.5);
        for (int ks = 1; ks < z; ks++) {
            const int ks2 = ks * 2;
            const double s1v = osc[ks2], c1v = osc[ks2 + 1];
            osc[ks2] = s2v; osc[ks2 + 1] = c2v;
            const double x = kPi * (ks - z);
            const double v0 = win[ks - 1] / ((x1 - x2) * x);
            kb1[ks - 1] = (x2 * s2v - x1 * s1v + (c2v - c1v) / x) * v0;
            kb2[ks - 1] = (s2v - s1v) * v0;
            s2v = coeff * s2v - osc[ks2 - 2];
            c2v = coeff * c2v - osc[ks2 - 1];
        }
        kb1[z - 1] = (x2 * x2 - x1 * x1) / (x1 - x2) * 0.5;
        kb2[z - 1] = -1.0;
    }
    int filterLength() const { return z2 - 1; }
    int filterLatency() const { return z - 1; }
    void build(const double* gains, double* out) const
    {
        const double* kb1 = k1.data(); const double* kb2 = k2.data();
        double x1 = 0.0, y1 = gains[0], x2 = center[1], y2 = gains[1];
        { const double c = y1 - y2, d = x1 * y2 - x2 * y1; for (int ks = 0; ks < z; ks++) out[ks] = c * kb1[ks] + d * kb2[ks]; }
        kb1 += zi; kb2 += zi; x1 = x2; y1 = y2;
        for (int i = 2; i < bands; i++) {
            x2 = center[(size_t)i]; y2 = gains[i];
            const double c = y1 - y2, d = x1 * y2 - x2 * y1;
            for (int ks = 0; ks < z; ks++) out[ks] += c * kb1[ks] + d * kb2[ks];
            kb1 += zi; kb2 += zi; x1 = x2; y1 = y2;
        }
        if (lastVirtual) { const double c = y1 - y2, d = x1 * y2 - y1; for (int ks = 0; ks < z; ks++) out[ks] += c * kb1[ks] + d * kb2[ks]; }
        for (int i = 0; i < z - 1; i++) out[z + i] = out[z - 2 - i];
    }
};

// ---- fractional-delay filter bank (avir.h: CDSPFracFilterBankLin<float>)
struct FilterBank {
    double wfLen2 = 0.0, wfFreq = 0.0, wfAlpha = 0.0;
    int fracCount = 0, order = -1, srcLen = 0, len = 0, size = 0;
    bool initRequired = false, srcBuilt = false;
    DesignedFilter ext;
    std::vector<double> src; std::vector<float> table; std::vector<uint8_t> fill;

    void copyInitParams(const FilterBank& s)
    {
        wfLen2 = s.wfLen2; wfFreq = s.wfFreq; wfAlpha = s.wfAlpha; fracCount = s.fracCount; order = s.order; srcLen = s.srcLen; len = s.len; size = s.size;
        srcBuilt = false; ext = s.ext;
        fill.assign(s.fill.size(), 0);
        for (size_t i = 0; i < fill.size(); i++) fill[i] = (uint8_t)(s.fill[i] << 2);
    }
    bool sameAs(const FilterBank& s) const { return order == s.order && wfLen2 == s.wfLen2 && wfFreq == s.wfFreq && wfAlpha == s.wfAlpha && fracCount == s.fracCount && ext.sameDesign(s.ext); }
    void init(int reqFrac, int reqOrder, double baseLen, double cutoff, double alpha, const DesignedFilter& e)
    {
        const double l2 = 0.5 * baseLen * reqFrac, fr = kPi * cutoff / reqFrac;
        if (reqOrder == order && l2 == wfLen2 && fr == wfFreq && alpha == wfAlpha && reqFrac == fracCount && e.sameDesign(ext)) { initRequired = false; return; }
        wfLen2 = l2; wfFreq = fr; wfAlpha = alpha; fracCount = reqFrac; order = reqOrder; ext = e;
        const LowPass p(wfLen2, wfFreq, wfAlpha);
        srcLen = (p.fl2 / reqFrac + 1) * 2;
        len = srcLen; if (!ext.taps.empty()) len += (int)ext.taps.size() - 1;
        len = (len + kAlign - 1) & ~(kAlign - 1);
        size = len * (reqOrder + 1);
        srcBuilt = false; initRequired = true;
    }
    void buildSource()
    {
        srcBuilt = true; initRequired = false;
        const LowPass p(wfLen2, wfFreq, wfAlpha);
        const int bufLen = srcLen * fracCount + 1, bufCenter = srcLen * fracCount / 2;     // InterpPoints = 2: BufOffs = 0
        std::vector<double> buf((size_t)bufLen, 0.0);
        p.generate(&buf[(size_t)(bufCenter - p.fl2)], fracCount);
        src.assign((size_t)((fracCount + 1) * srcLen), 0.0);
        fill.assign((size_t)(fracCount + 1), 0);
        double* op = src.data();
        for (int i = fracCount; i >= 0; i--) { const double* q = buf.data() + i; for (int j = 0; j < srcLen; j++) { *op++ = *q; q += fracCount; } }
        table.assign((size_t)((fracCount + 1) * size), 0.0f);
    }
    void create(int k)
    {
        if (fill[(size_t)k] != 0) return;
        fill[(size_t)k] |= 1;
        const int extLen = (int)ext.taps.size();
        const int resLatency = extLen / 2 + srcLen / 2;
        int resLen = srcLen; if (extLen > 0) resLen += extLen - 1;
        const int resOffs = len / 2 - resLatency;
        float* op = &table[(size_t)(k * size)];
        for (int i = 0; i < resOffs; i++) op[i] = 0.0f;
        for (int i = resOffs + resLen; i < len; i++) op[i] = 0.0f;
        op += resOffs;
        const double* sf = &src[(size_t)(k * srcLen)];
        if (extLen == 0) { for (int i = 0; i < resLen; i++) op[i] = (float)sf[i]; return; }
        const double* ef = ext.taps.data();
        for (int j = 0; j < resLen; j++) {
            int kk = 0, l = j - extLen + 1, r = l + extLen;
            if (l < 0) { kk -= l; l = 0; }
            if (r > srcLen) r = srcLen;
            const double* eb = ef + kk; const double* sb = sf + l;
            double s = 0.0; const int n = r - l;
            for (int i = 0; i < n; i++) s += eb[i] * sb[i];
            op[j] = (float)s;
        }
    }
    const float* filter(int i)
    {
        if (!srcBuilt) buildSource();
        float* res = &table[(size_t)(i * size)];
        if ((fill[(size_t)i] & 2) == 0) {
            create(i); fill[(size_t)i] |= 2;
            if (order > 0) { create(i + 1); const float* r2 = res + size; float* op = res + len; for (int j = 0; j < len; j++) op[j] = r2[j] - res[j]; }
        }
        return res;
    }
    int initComplexity(const std::vector<uint8_t>& used) const
    {
        const int useCost = len * order + srcLen * (int)ext.taps.size();
        int ic;
        if (initRequired) { ic = fracCount * srcLen * 65; for (int i = 0; i < fracCount; i++) ic += used[(size_t)i] ? useCost : 0; }
        else { ic = 0; for (int i = 0; i < fracCount; i++) if (used[(size_t)i] != 0) ic += (fill[(size_t)i] == 0 ? useCost : 0); }
        return ic;
    }
};

// ---- one filtering step
struct RPos { int srcPosInt, fti; const float* ftp; float x; int srcOffs; };
struct Step {
    bool upsample = false; int factor = 0;            // factor 0: the resizing (interpolation) step
    std::vector<float> flt; int fltCap = 0;            // taps (padded with zeros to a multiple of 8); fltCap also in model mode
    DesignedFilter orig; int origCap = 0;              // the designed filter kept for the filter bank (combined modes)
    double dcGain = 1.0; int latency = 0;
    std::vector<float> prefixDC, suffixDC; int prefixCap = 0, suffixCap = 0;
    int edge = 0;
    int inLen = 0, inPrefix = 0, inSuffix = 0, outLen = 0, outPrefix = 0, outSuffix = 0;
    FilterBank* bank = nullptr; std::vector<RPos> rpos;
};

int alignedCap(int req) { return (req + kAlign - 1) & ~(kAlign - 1); }

struct Resizer {
    FilterBank fixedBank;
    Resizer() { initBank(fixedBank, 1.0, false, DesignedFilter()); for (int i = 0; i < fixedBank.fracCount; i++) fixedBank.filter(i); }

    static void initBank(FilterBank& b, double cutoffMult, bool hiOrder, const DesignedFilter& ext)
    {
        const double snr = -6.02 * (8 + 3);
        int order, frac;
        if (hiOrder) { order = 1; frac = (int)std::ceil(0.23134052 * std::exp(-0.058062929 * snr)); }
        else { order = 0; frac = (int)std::ceil(0.33287686 * std::exp(-0.11334583 * snr)); }
        if (frac < 2) frac = 2;
        b.init(frac, order, kIntFltLen / cutoffMult, kIntFltCutoff * cutoffMult, kIntFltAlpha, ext);
    }

    // avir.h: assignFilterParams
    static void assignFilter(Step& fs, bool isUp, int factor, double cutoff, double dcGain, bool keepOrig, bool model)
    {
        double alpha, len2, freq;
        if (cutoff == 0.0) { const double m = 2.0 / factor; alpha = kHBFltAlpha; len2 = 0.5 * kHBFltLen / m; freq = kPi * kHBFltCutoff * m; }
        else { alpha = kLPFltAlpha; len2 = 0.25 * kLPFltBaseLen / cutoff; freq = kPi * kLPFltCutoffMult * cutoff; }
        if (isUp) { len2 *= factor; freq /= factor; fs.dcGain = dcGain * factor; } else fs.dcGain = dcGain;
        fs.orig.len2 = len2; fs.orig.freq = freq; fs.orig.alpha = alpha; fs.orig.dcGain = fs.dcGain;
        const LowPass w(len2, freq, alpha);
        fs.upsample = isUp; fs.factor = factor; fs.latency = w.fl2;
        int ext;
        if (model) {
            fs.fltCap = alignedCap(w.len); ext = fs.fltCap - w.len;
            if (keepOrig) { fs.orig.taps.assign((size_t)w.len, 0.0); fs.origCap = w.len; }
        } else {
            fs.orig.taps.assign((size_t)w.len, 0.0);
            w.generate(fs.orig.taps.data(), 1.0);
            trimFilter(fs.orig.taps, fs.latency);
            normalize(fs.orig.taps.data(), (int)fs.orig.taps.size(), fs.dcGain);
            const int n = (int)fs.orig.taps.size();
            fs.fltCap = alignedCap(n); ext = fs.fltCap - n;
            fs.flt.assign((size_t)fs.fltCap, 0.0f);
            for (int i = 0; i < n; i++) fs.flt[(size_t)i] = (float)fs.orig.taps[(size_t)i];
            fs.origCap = n;
            if (!keepOrig) { fs.orig.taps.clear(); fs.origCap = 0; }
        }
        if (isUp) {
            int l = fs.fltCap - fs.latency - factor - ext;
            fs.prefixCap = alignedCap(l); fs.suffixCap = alignedCap(fs.latency);
            if (model) return;
            fs.prefixDC.assign((size_t)fs.prefixCap, 0.0f); fs.suffixDC.assign((size_t)fs.suffixCap, 0.0f);
            const float* ip = &fs.flt[(size_t)(fs.latency + factor)];
            for (int i = 0; i < l; i++) fs.prefixDC[(size_t)i] = ip[i];
            for (;;) { ip += factor; l -= factor; if (l <= 0) break; for (int i = 0; i < l; i++) fs.prefixDC[(size_t)i] += ip[i]; }
            l = fs.latency;
            float* op = fs.suffixDC.data();
            for (int i = 0; i < l; i++) op[i] = fs.flt[(size_t)i];
            for (;;) { op += factor; l -= factor; if (l <= 0) break; for (int i = 0; i < l; i++) op[i] += fs.flt[(size_t)i]; }
        } else if (!keepOrig) fs.edge = kEdgePixels;
    }

    // avir.h: addCorrectionFilter
    void addCorrection(std::vector<std::unique_ptr<Step>>& steps, double bw, bool pre, bool model)
    {
        if (!pre) steps.emplace_back(new Step());
        Step& fs = pre ? *steps[0] : *steps.back();
        fs.upsample = false; fs.factor = 1; fs.dcGain = 1.0; fs.edge = pre ? kEdgePixels : 0;
        if (model) { const int l = (int)std::ceil(kCorrFltLen * 0.5); fs.latency = l - 1; fs.fltCap = alignedCap(l * 2 - 1); return; }
        constexpr int kBins = 65;
        double bins[kBins]; for (double& b : bins) b = 1.0;
        double curbw = 1.0, re, im;
        const int si = pre ? 1 : 0;
        for (int i = si; i < (int)steps.size() - (si ^ 1); i++) {
            Step& s = *steps[(size_t)i];
            if (s.upsample) { curbw *= s.factor; if (s.origCap > 0) continue; }
            const double dcg = 1.0 / s.dcGain;
            const float* f; int fl;
            if (s.factor == 0) { f = s.bank->filter(0); fl = s.bank->len; } else { f = s.flt.data(); fl = s.fltCap; }
            for (int j = 0; j < kBins; j++) {
                const double th = kPi * bw / curbw * j / (kBins - 1);
                response(f, fl, th, re, im);
                bins[j] /= std::sqrt(re * re + im * im) * dcg;
            }
            if (!s.upsample && s.factor > 1) curbw /= s.factor;
        }
        Equalizer eq; eq.init(bw * 2.0, kCorrFltLen, kBins, bw, kCorrFltAlpha);
        fs.latency = eq.filterLatency();
        std::vector<double> filter((size_t)eq.filterLength());
        eq.build(bins, filter.data());
        normalize(filter.data(), (int)filter.size(), 1.0);
        trimFilter(filter, fs.latency);
        normalize(filter.data(), (int)filter.size(), 1.0);
        fs.fltCap = alignedCap((int)filter.size());
        fs.flt.assign((size_t)fs.fltCap, 0.0f);
        for (size_t i = 0; i < filter.size(); i++) fs.flt[i] = (float)filter[i];
    }

    // avir.h: buildFilterSteps (mode bit 0: filter and interpolator combined, bit 1: 1st-order interpolation; no half-band steps in modes 0..3)
    void buildSteps(std::vector<std::unique_ptr<Step>>& steps, double k, int& resizeStep, FilterBank& bank, double dcGain, int mode, bool model)
    {
        steps.clear();
        const bool combo = (mode & 1) != 0, hiOrder = (mode & 2) != 0;
        const double bw = 1.0 / k;
        const int upFactor = ((int)std::floor(k) < 2 ? 2 : 1);
        double intCutoffMult, fltCutoff, corrbw; bool pre;
        Step* reuse = nullptr; Step* extStep = nullptr;
        if (k <= 1.0) { pre = true; fltCutoff = 1.0; corrbw = 1.0; steps.emplace_back(new Step()); }
        else { pre = false; fltCutoff = bw; corrbw = bw; }
        if (upFactor > 1) {
            steps.emplace_back(new Step()); Step& fs = *steps.back();
            assignFilter(fs, true, upFactor, fltCutoff, dcGain, combo, model);
            intCutoffMult = fltCutoff * 2.0 / upFactor;
            extStep = combo ? &fs : nullptr;
        } else {
            int down;
            for (;;) {
                down = (int)std::floor(0.5 / fltCutoff);
                bool halfband;
                if (down > 16) { halfband = true; down = 16; } else halfband = false;    // (UseHalfband is mode bit 2: never set here)
                if (halfband) { steps.emplace_back(new Step()); assignFilter(*steps.back(), false, down, 0.0, 1.0, false, model); fltCutoff *= down; }
                else { if (down < 1) down = 1; break; }
            }
            steps.emplace_back(new Step()); Step& fs = *steps.back();
            assignFilter(fs, false, down, fltCutoff, dcGain, combo, model);
            intCutoffMult = fltCutoff / 0.5;
            if (combo) { reuse = &fs; extStep = &fs; } else intCutoffMult *= down;
        }
        if (!reuse) steps.emplace_back(new Step());
        Step& fs = reuse ? *reuse : *steps.back();
        resizeStep = (int)steps.size() - 1;
        fs.upsample = false; fs.factor = 0;
        fs.dcGain = extStep ? extStep->dcGain : 1.0;
        initBank(bank, intCutoffMult, hiOrder, extStep ? extStep->orig : fs.orig);
        fs.bank = bank.sameAs(fixedBank) ? &fixedBank : &bank;
        addCorrection(steps, corrbw, pre, model);
    }

    // avir.h: updateFilterStepBuffers + fillRPosBuf + extendUpsample (k, o are modified step by step)
    static void layout(std::vector<std::unique_ptr<Step>>& steps, double k, double o, int srcLen, int newLen)
    {
        int up = -1;
        for (size_t i = 0; i < steps.size(); i++) {
            Step& fs = *steps[i];
            fs.inLen = srcLen;
            if (fs.upsample) {
                up = (int)i; k *= fs.factor; o *= fs.factor;
                fs.inPrefix = 0; fs.inSuffix = 0;
                fs.outLen = fs.inLen * fs.factor; fs.outPrefix = fs.latency; fs.outSuffix = fs.fltCap - fs.latency - fs.factor;
                int l0 = fs.outPrefix + fs.outLen + fs.outSuffix;
                const int l = fs.inLen * fs.factor + fs.suffixCap;
                if (l > l0) fs.outSuffix += l - l0;
                l0 = fs.outLen + fs.outSuffix;
                if (fs.prefixCap > l0) fs.outSuffix += fs.prefixCap - l0;
            } else if (fs.factor == 0) {
                const int d2 = fs.bank->len / 2, d21 = d2 - 1;
                const int lpix = (int)std::floor(o) - d21;
                fs.inPrefix = lpix < 0 ? -lpix : 0;
                const int rpix = (int)std::floor(o + (newLen - 1) * k) + d2 + 1;
                fs.inSuffix = rpix > fs.inLen ? rpix - fs.inLen : 0;
                fs.outLen = newLen;
                fs.rpos.resize((size_t)newLen);
                const int fc = fs.bank->fracCount;
                for (int j = 0; j < newLen; j++) {
                    const double sp = o + k * j;
                    const int spi = (int)std::floor(sp);
                    const double x = (sp - spi) * fc;
                    const int fti = (int)x;
                    fs.rpos[(size_t)j] = { spi, fti, nullptr, (float)(x - fti), 0 };
                }
            } else {
                k /= fs.factor; o /= fs.factor; o += fs.edge;
                fs.inPrefix = fs.latency; fs.inSuffix = fs.fltCap - fs.latency - 1;
                fs.outLen = (fs.inLen + fs.factor - 1) / fs.factor + fs.edge;
                fs.inSuffix += (fs.outLen - 1) * fs.factor + 1 - fs.inLen;
                fs.inPrefix += fs.edge * fs.factor;
                fs.outLen += fs.edge;
            }
            srcLen = fs.outLen;
        }
        if (up != -1) {
            Step& fs = *steps[(size_t)up]; Step& nx = *steps[(size_t)up + 1];
            fs.inPrefix = (nx.inPrefix + fs.factor - 1) / fs.factor; fs.outPrefix += fs.inPrefix * fs.factor; nx.inPrefix = 0;
            fs.inSuffix = (nx.inSuffix + fs.factor - 1) / fs.factor; fs.outSuffix += fs.inSuffix * fs.factor; nx.inSuffix = 0;
        }
    }

    // avir.h: calcComplexity (de-interleaved processing: packmode 1)
    static int complexity(const std::vector<std::unique_ptr<Step>>& steps, const std::vector<uint8_t>& used, int lines)
    {
        int s = 0, s2 = 0;
        for (const auto& sp : steps) {
            const Step& fs = *sp;
            s2 += 65 * fs.fltCap;
            if (fs.upsample) {
                if (fs.origCap > 0) continue;
                s += (fs.fltCap * (fs.inPrefix + fs.inLen + fs.inSuffix) + fs.suffixCap + fs.prefixCap) * kChannels;
            } else if (fs.factor == 0) {
                s += fs.bank->len * (fs.bank->order + kChannels) * fs.outLen;
                s2 += fs.bank->initComplexity(used);
            } else s += fs.fltCap * kChannels * fs.outLen;
        }
        return s + s2 / lines;
    }

    // picks the build mode as resizeImage does: all four are laid out as models (lengths only, no filter design) and costed
    int plan(const FilterBank* previousBank, double k, double o, int srcLen, int newLen, int lines)
    {
        int best = 0x7FFFFFFF, use = 1;
        for (int m = 0; m < 4; m++) {
            FilterBank tmp; if (previousBank) tmp.copyInitParams(*previousBank);
            std::vector<std::unique_ptr<Step>> ts; int rs = 0;
            buildSteps(ts, k, rs, tmp, 1.0, m, true);
            layout(ts, k, o, srcLen, newLen);
            const Step& r = *ts[(size_t)rs];
            std::vector<uint8_t> used((size_t)r.bank->fracCount, 0);
            for (const RPos& p : r.rpos) used[(size_t)p.fti] |= 1;
            const int c = complexity(ts, used, lines);
            if (c < best) { use = m; best = c; }
        }
        return use;
    }
};

// binary32 dot product of a filter with a signal in the order of the float8 code: eight lane sums, then ((s0+s4)+(s1+s5))+((s2+s6)+(s3+s7))
inline float dot8(const float* f, const float* x, int len)
{
    float s[8];
    for (int j = 0; j < 8; j++) s[j] = f[j] * x[j];
    for (int i = 8; i < len; i += 8) for (int j = 0; j < 8; j++) s[j] += f[i + j] * x[i + j];
    return ((s[0] + s[4]) + (s[1] + s[5])) + ((s[2] + s[6]) + (s[3] + s[7]));
}

// one scanline of one channel through all steps; `line` holds the input samples, the result (outLen of the last step) is returned in `out`
struct Workspace { std::vector<float> a, b, blend; };
void runSteps(const std::vector<std::unique_ptr<Step>>& steps, const float* in, int inLen, float* out, int outStride, Workspace& ws)
{
    // buffers with generous margins; `cur` points at sample 0 of the current signal
    size_t need = 64;
    for (const auto& sp : steps) need = std::max(need, (size_t)(sp->inPrefix + sp->inLen + sp->inSuffix + sp->outPrefix + sp->outLen + sp->outSuffix + 4 * kAlign + 64));
    if (ws.a.size() < 2 * need) { ws.a.assign(2 * need, 0.0f); ws.b.assign(2 * need, 0.0f); }
    float* cur = ws.a.data() + need; float* nxt = ws.b.data() + need;
    std::memcpy(cur, in, (size_t)inLen * sizeof(float));
    for (size_t si = 0; si < steps.size(); si++) {
        const Step& fs = *steps[si];
        const bool lastStep = si + 1 == steps.size();
        if (!fs.upsample && fs.inPrefix + fs.inSuffix != 0) {           // prepareInBuf: the edges are replicated
            for (int i = 1; i <= fs.inPrefix; i++) cur[-i] = cur[0];
            for (int i = 0; i < fs.inSuffix; i++) cur[fs.inLen + i] = cur[fs.inLen - 1];
        }
        if (fs.upsample) {
            float* op0 = nxt - fs.outPrefix;
            std::memset(op0, 0, (size_t)(fs.outPrefix + fs.outLen + fs.outSuffix) * sizeof(float));
            const float* ip = cur;
            if (fs.origCap > 0) {                                       // combined modes: zero stuffing only (the interpolation filters carry the low-pass filter)
                op0 += fs.outPrefix % fs.factor;
                for (int l = fs.outPrefix / fs.factor; l > 0; l--) { op0[0] = ip[0]; op0 += fs.factor; }
                for (int l = fs.inLen - 1; l > 0; l--) { op0[0] = ip[0]; op0 += fs.factor; ip++; }
                for (int l = fs.outSuffix / fs.factor; l >= 0; l--) { op0[0] = ip[0]; op0 += fs.factor; }
            } else {
                const float* f = fs.flt.data(); const int flen = fs.fltCap;
                float v = ip[0];
                for (int l = fs.inPrefix; l > 0; l--) { for (int i = 0; i < flen; i++) op0[i] = op0[i] + f[i] * v; op0 += fs.factor; }
                for (int l = fs.inLen - 1; l > 0; l--) { v = ip[0]; for (int i = 0; i < flen; i++) op0[i] = op0[i] + f[i] * v; ip++; op0 += fs.factor; }
                v = ip[0];
                for (int l = fs.inSuffix; l >= 0; l--) { for (int i = 0; i < flen; i++) op0[i] = op0[i] + f[i] * v; op0 += fs.factor; }
                for (int i = 0; i < fs.suffixCap; i++) op0[i] = op0[i] + fs.suffixDC[(size_t)i] * v;
                v = cur[0];
                op0 = nxt - fs.inPrefix * fs.factor;
                for (int i = 0; i < fs.prefixCap; i++) op0[i] = op0[i] + fs.prefixDC[(size_t)i] * v;
            }
        } else if (fs.factor != 0) {
            const float* ip = cur - fs.edge * fs.factor - fs.latency;
            for (int l = 0; l < fs.outLen; l++) {
                const float r = dot8(fs.flt.data(), ip, fs.fltCap);
                if (lastStep) out[(size_t)l * outStride] = r; else nxt[l] = r;
                ip += fs.factor;
            }
        } else {
            const int flen = fs.bank->len;
            for (int l = 0; l < fs.outLen; l++) {
                const RPos& rp = fs.rpos[(size_t)l];
                const float* src = cur + rp.srcOffs;
                float r;
                if (fs.bank->order == 1) {
                    if (ws.blend.size() < (size_t)flen) ws.blend.resize((size_t)flen);
                    float* xx = ws.blend.data(); const float* f2 = rp.ftp + flen;
                    for (int i = 0; i < flen; i++) xx[i] = rp.ftp[i] + f2[i] * rp.x;
                    r = dot8(xx, src, flen);
                } else r = dot8(rp.ftp, src, flen);
                if (lastStep) out[(size_t)l * outStride] = r; else nxt[l] = r;
            }
        }
        std::swap(cur, nxt);
    }
}

} // namespace

std::vector<uint8_t> resizeSquare(const uint8_t* rgba, unsigned from, unsigned to)
{
    if (from == 0 || to == 0) throw std::runtime_error("resizeSquare: empty image");
    const int srcN = (int)from, newN = (int)to;
    Resizer rz;
    double k, o = 0.0;
    if (newN > srcN) k = (double)(srcN - 1) / (newN - 1); else { k = (double)srcN / newN; o += (k - 1.0) * 0.5; }

    // ---- horizontal pass: every source row to newN samples per channel (binary32 planes)
    std::vector<std::unique_ptr<Step>> steps; int resizeStep = 0;
    FilterBank bankH;
    const int modeH = rz.plan(nullptr, k, o, srcN, newN, srcN);
    rz.buildSteps(steps, k, resizeStep, bankH, 1.0, modeH, false);
    Resizer::layout(steps, k, o, srcN, newN);
    auto bindFilters = [&](std::vector<std::unique_ptr<Step>>& st, int rs) {
        Step& fs = *st[(size_t)rs];
        const int d21 = fs.bank->len / 2 - 1;
        for (RPos& rp : fs.rpos) { rp.ftp = fs.bank->filter(rp.fti); rp.srcOffs = rp.srcPosInt - d21; }
    };
    bindFilters(steps, resizeStep);
    std::vector<float> mid((size_t)srcN * newN * kChannels);           // [row][channel][x]
    // rows (then columns) are independent and every filter the steps read has been built above: the scanlines are spread over threads,
    // each with its own scratch buffers -- the same bytes whatever the thread count
    const unsigned hw = std::thread::hardware_concurrency();
    const int workers = (int)std::min<size_t>(hw ? (hw < 16 ? hw : 16) : 1, std::max<size_t>(1, ((size_t)srcN * (size_t)std::max(srcN, newN)) >> 16));
    auto spread = [&](int count, auto&& body) {
        if (workers <= 1 || count < 2 * workers) { Workspace ws; body(0, count, ws); return; }
        std::vector<std::thread> pool;
        for (int t = 0; t < workers; t++)
            pool.emplace_back([&, t]() { Workspace ws; body((int)((long long)count * t / workers), (int)((long long)count * (t + 1) / workers), ws); });
        for (auto& th : pool) th.join();
    };
    spread(srcN, [&](int y0, int y1, Workspace& ws) {
        std::vector<float> line((size_t)srcN);
        for (int y = y0; y < y1; y++)
            for (int c = 0; c < kChannels; c++) {
                for (int x = 0; x < srcN; x++) line[(size_t)x] = (float)rgba[((size_t)y * srcN + (size_t)x) * 4 + (size_t)c];
                runSteps(steps, line.data(), srcN, &mid[((size_t)y * kChannels + (size_t)c) * newN], 1, ws);
            }
    });

    // ---- vertical pass: the mode is chosen again (the model sees the filters the horizontal pass created); same mode and same k: the steps are reused
    const int modeV = rz.plan(&bankH, k, o, srcN, newN, newN);
    if (modeV != modeH) rz.buildSteps(steps, k, resizeStep, bankH, 1.0, modeV, false);
    Resizer::layout(steps, k, o, srcN, newN);
    bindFilters(steps, resizeStep);
    std::vector<float> res((size_t)newN * newN * kChannels);            // [row][channel][x]
    spread(newN, [&](int x0, int x1, Workspace& ws) {
        std::vector<float> col((size_t)srcN);
        for (int x = x0; x < x1; x++)
            for (int c = 0; c < kChannels; c++) {
                for (int y = 0; y < srcN; y++) col[(size_t)y] = mid[((size_t)y * kChannels + (size_t)c) * newN + (size_t)x];
                runSteps(steps, col.data(), srcN, &res[(size_t)c * newN + (size_t)x], newN * kChannels, ws);
            }
    });

    // ---- output: round half to even, clamp to [0, 255] (CImageResizerDithererDefDIL with TrMul = 1), interleave
    std::vector<uint8_t> out((size_t)newN * newN * 4);
    for (int y = 0; y < newN; y++)
        for (int c = 0; c < kChannels; c++)
            for (int x = 0; x < newN; x++) {
                float v = std::nearbyintf(res[((size_t)y * kChannels + (size_t)c) * newN + (size_t)x]);
                v = v < 0.0f ? 0.0f : v; v = v > 255.0f ? 255.0f : v;
                out[((size_t)y * newN + (size_t)x) * 4 + (size_t)c] = (uint8_t)v;
            }
    return out;
}

} // namespace gmupt
