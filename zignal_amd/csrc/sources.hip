// Several MLS sources measured at once.
//
// zignal's MLS class is "limited to one channel of audio": a multichannel measurement is left to the caller, who makes
// one MLS per output (different taps of one order, "to minimise cross talk"), plays them together, and has "to keep
// track of which channel is paired with which MLS" when extracting (zignal/measure/mls.py:338-343). Extracting means
// one get_impulse per (recorded channel, source) pair - trim_and_avgerage + xcorr_circular with that source's sequence
// (mls.py:315-328) - so the capture is read, trimmed and averaged once per source.
//
// Here the sources are K plans of one order and the extraction is one call over the whole capture:
//   * the average over the periods does not depend on the taps, so the periods are added up ONCE (time order,
//     period_sum.hip / layout.hip) and every plan gathers, transforms and scatters that one period of sums - the capture
//     is read once instead of K times whenever that moves fewer bytes;
//   * with a de-emphasis filter the single sweep is the fused one of iir.hip (sums + period-boundary states, then the
//     filter over one period), again once for all sources;
//   * the plans run one after the other on one stream, so they share ONE set of scratch - the intermediate between the
//     passes and the layout staging of the first plan - instead of K.
// Response of source k, channel c, lag m:  d_h[k*source_stride + m*out_row_stride + c*out_ch_stride].
#include "zmls_internal.h"

using namespace zmls;

namespace {

constexpr int kMaxSources = 64;

int check_sources(zmls_plan *const *plans, int n_plans)
{
    if (!plans || n_plans < 1 || n_plans > kMaxSources) { set_error("between 1 and %d source plans", kMaxSources); return ZMLS_E_ARG; }
    for (int k = 0; k < n_plans; ++k) {
        if (!plans[k]) { set_error("source plan %d is null", k); return ZMLS_E_ARG; }
        if (int rc = sequence_only(plans[k])) return rc;
        if (plans[k]->N != plans[0]->N || plans[k]->dtype != plans[0]->dtype || plans[k]->device != plans[0]->device) {
            set_error("source plan %d differs from plan 0 in order, dtype or device: sources that play together share all three", k);
            return ZMLS_E_ARG;
        }
    }
    return ZMLS_OK;
}

// reading the capture once (periods + one period of sums written) against once per source, in period-sized units
bool shared_sums_pay(const zmls_plan *p0, int n_plans, int64_t periods)
{
    return p0->npass > 1 && (int64_t)(n_plans - 1) * (periods - 1) > 2;
}

} // namespace

extern "C" {

int zmls_sources_reserve(zmls_plan *const *plans, int n_plans, zmls_iir *f, int64_t n_rows, int64_t n_ch)
{
    int rc = check_sources(plans, n_plans);
    if (rc) return rc;
    if (n_ch < 1) { set_error("need at least one channel"); return ZMLS_E_ARG; }
    int64_t periods = 0;
    rc = capture_periods(plans[0], n_rows, &periods);
    if (rc) return rc;
    if (f) return zmls_iir_reserve_fused(f, plans[0], n_rows, n_ch);        // includes the plan's workspace and staging
    return zmls_plan_reserve_staging(plans[0], n_ch);                       // includes the workspace
}

int zmls_impulse_sources_device(zmls_plan *const *plans, int n_plans, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows,
                                int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h, int64_t source_stride,
                                int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    int rc = check_sources(plans, n_plans);
    if (rc) return rc;
    if (!d_x || !d_h) { set_error("null argument"); return ZMLS_E_ARG; }
    zmls_plan *p0 = plans[0];
    const size_t is = dtype_size(in_dtype), ts = dtype_size(p0->dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1 || row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1 || (source_stride < 1 && n_plans > 1)) {
        set_error("bad shape/strides");
        return ZMLS_E_ARG;
    }
    int64_t periods = 0;
    rc = capture_periods(p0, n_rows, &periods);
    if (rc) return rc;
    ZDEVICE(p0->device);
    hipStream_t st = (hipStream_t)stream;
    const Scratch sc = {p0->d_work, p0->work_bytes, {p0->d_stage[0], p0->d_stage[1]}, p0->stage_bytes};
    auto response = [&](int k, int64_t c0) { return (void *)((char *)d_h + ((size_t)k * (size_t)source_stride + (size_t)(c0 * out_ch_stride)) * ts); };

    if (f) {
        const double *sums = nullptr;
        int64_t srs = 0, scs = 0;
        rc = filtered_period_sums(p0, f, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, &sums, &srs, &scs, st);
        for (int k = 0; k < n_plans && !rc; ++k)
            rc = impulse_device_core(plans[k], sums, ZMLS_F64, 1, 0, periods, n_ch, srs, scs, response(k, 0), out_row_stride, out_ch_stride, st,
                                     nullptr, nullptr, 0, &sc);
        return rc;
    }

    const int64_t E = (int64_t)(kVecBytes / ts);
    const int64_t cc = (int64_t)(sc.stage_bytes / ((size_t)p0->L * ts)) / E * E;          // channels a staging array holds
    if (!shared_sums_pay(p0, n_plans, periods) || cc < E) {
        for (int k = 0; k < n_plans && !rc; ++k)
            rc = impulse_device_core(plans[k], d_x, in_dtype, periods, 1, periods, n_ch, row_stride, ch_stride, response(k, 0), out_row_stride,
                                     out_ch_stride, st, nullptr, nullptr, 0, &sc);
        return rc;
    }
    // the periods added up once per channel chunk into the first staging array (frames x channels, padded to whole
    // vectors with zero columns), then every source's passes on it; responses that are contiguous in time leave
    // through the second staging array inside impulse_device_core
    const bool time_contiguous = row_stride == 1 && (n_ch == 1 || ch_stride != 1);
    for (int64_t c0 = 0; c0 < n_ch && !rc; c0 += cc) {
        const int64_t n = (n_ch - c0 < cc) ? n_ch - c0 : cc;
        const int64_t Cs = (n + E - 1) / E * E;
        if (time_contiguous) rc = launch_sum_transpose(d_x, in_dtype, p0->dtype, p0->L, periods, 1, ch_stride, c0, n_ch, sc.stage[0], Cs, st);
        else rc = launch_period_sum(d_x, in_dtype, p0->dtype, p0->L, periods, 1, row_stride, ch_stride, 1, 0, c0, Cs, n_ch, sc.stage[0], st);
        const Scratch rest = {sc.work, sc.work_bytes, {nullptr, sc.stage[1]}, sc.stage_bytes};
        for (int k = 0; k < n_plans && !rc; ++k)
            rc = staged_sums_to_responses(plans[k], rest, sc.stage[0], Cs, periods, n, response(k, c0), out_row_stride, out_ch_stride, st);
    }
    return rc;
}

} // extern "C"
