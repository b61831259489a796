// merge.cpp — mergePairs() for one sample (/root/reference/R/paired.R:92-201): the step after dada()
// on paired reads.  The bookkeeping of the R function (unique forward/reverse pairs, abundances, order) and the two C
// helpers it calls (C_eval_pair / C_pair_consensus, /root/reference/src/evaluate.cpp:73-174) are host code; every
// forward x reverse-complement alignment (R's nwalign(x, y, band=-1): C_nwalign -> nwalign_endsfree, unbanded,
// evaluate.cpp:18-62) runs on the device through dada2hip_nwvec.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dada2hip.h"

struct dada2hip_mergers {
  std::vector<std::string> sequence;
  std::vector<int32_t> abundance, forward, reverse, nmatch, nmismatch, nindel, prefer, accept;
};

namespace {

void set_err(char *errbuf, size_t errlen, const char *m) {
  if (errbuf && errlen) snprintf(errbuf, errlen, "%s", m);
}

std::string revcomp(const char *s) {   // R/misc.R rc()
  const size_t n = strlen(s);
  std::string r(n, 'N');
  for (size_t i = 0; i < n; i++) {
    const char c = s[n - 1 - i];
    r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
  }
  return r;
}

// C_eval_pair (evaluate.cpp:73-114): the internal part of the alignment is what lies between the end gaps of either string
void eval_pair(const std::string &s1, const std::string &s2, int32_t &match, int32_t &mismatch, int32_t &indel) {
  const int n = (int)s1.size();
  bool g1 = true, g2 = true;
  int start = -1;
  do {
    start++;
    g1 = g1 && start < n && s1[start] == '-';
    g2 = g2 && start < n && s2[start] == '-';
  } while ((g1 || g2) && start < n);
  g1 = g2 = true;
  int end = n;
  do {
    end--;
    if (end < 0) break;
    g1 = g1 && s1[end] == '-';
    g2 = g2 && s2[end] == '-';
  } while ((g1 || g2) && end >= start);
  match = mismatch = indel = 0;
  for (int i = start; i <= end; i++) {
    if (s1[i] == '-' || s2[i] == '-') indel++;
    else if (s1[i] == s2[i]) match++;
    else mismatch++;
  }
}

// C_pair_consensus (evaluate.cpp:124-174)
std::string pair_consensus(const std::string &s1, const std::string &s2, int prefer, bool trim_overhang) {
  const int n = (int)s1.size();
  std::string o(n, '-');
  for (int i = 0; i < n; i++) {
    if (s1[i] == s2[i]) o[i] = s1[i];
    else if (s2[i] == '-') o[i] = s1[i];
    else if (s1[i] == '-') o[i] = s2[i];
    else o[i] = prefer == 1 ? s1[i] : (prefer == 2 ? s2[i] : 'N');
  }
  if (trim_overhang) {
    for (int i = 0; i < n; i++) { if (s1[i] != '-') break; o[i] = '-'; }
    for (int i = n - 1; i >= 0; i--) { if (s2[i] != '-') break; o[i] = '-'; }
  }
  o.erase(std::remove(o.begin(), o.end(), '-'), o.end());
  return o;
}

}  // namespace

extern "C" {

static int merge_pairs_body(int64_t nreads, const int32_t *fwd, const int32_t *rev, int32_t nF, const char *const *seqsF,
                            const int32_t *n0F, int32_t nR, const char *const *seqsR, const int32_t *n0R, int32_t min_overlap,
                            int32_t max_mismatch, int32_t trim_overhang, int32_t just_concatenate, int32_t device,
                            dada2hip_mergers **out, char *errbuf, size_t errlen) {
  if (out) *out = nullptr;
  if (!out || nreads < 0 || (nreads > 0 && (!fwd || !rev)) || !seqsF || !seqsR || !n0F || !n0R) {
    set_err(errbuf, errlen, "dada2hip: merge_pairs needs the two read maps and both clustering tables.");
    return DADA2HIP_ERR_INPUT;
  }
  // unique(pairdf) in order of first appearance, NA pairs dropped (paired.R:122-125); abundance = table() count (:172-173)
  std::map<std::pair<int32_t, int32_t>, int32_t> index;
  std::vector<std::pair<int32_t, int32_t>> ups;
  std::vector<int32_t> abund;
  for (int64_t i = 0; i < nreads; i++) {
    const int32_t f = fwd[i], r = rev[i];
    if (f == DADA2HIP_NA_INTEGER || r == DADA2HIP_NA_INTEGER) continue;
    if (f < 1 || f > nF || r < 1 || r > nR) {
      set_err(errbuf, errlen, "Non-corresponding derep-class and dada-class objects.");   // paired.R:116
      return DADA2HIP_ERR_INPUT;
    }
    auto it = index.find({f, r});
    if (it == index.end()) { index.emplace(std::make_pair(f, r), (int32_t)ups.size()); ups.push_back({f, r}); abund.push_back(1); }
    else abund[it->second]++;
  }
  const size_t P = ups.size();
  std::unique_ptr<dada2hip_mergers> m(new dada2hip_mergers());
  std::vector<std::string> seq(P);
  std::vector<int32_t> nmatch(P, 0), nmismatch(P, 0), nindel(P, 0), prefer(P, DADA2HIP_NA_INTEGER), accept(P, 1);
  if (P > 0) {
    std::vector<std::string> R(P);
    for (size_t p = 0; p < P; p++) R[p] = revcomp(seqsR[ups[p].second - 1]);
    if (just_concatenate) {                                                        // paired.R:139-147
      for (size_t p = 0; p < P; p++) seq[p] = std::string(seqsF[ups[p].first - 1]) + "NNNNNNNNNN" + R[p];
    } else {
      // mismatches and gaps are penalised heavily so that zero-mismatch merges win (paired.R:152-157)
      const int32_t match = 1, mismatch = max_mismatch == 0 ? -64 : -8, gap = mismatch;
      std::vector<const char *> s1(P), s2(P);
      std::vector<std::vector<char>> bufs(2 * P);
      std::vector<char *> outs(2 * P);
      for (size_t p = 0; p < P; p++) {
        s1[p] = seqsF[ups[p].first - 1];
        s2[p] = R[p].c_str();
        const size_t cap = strlen(s1[p]) + R[p].size() + 1;
        bufs[2 * p].assign(cap, 0); bufs[2 * p + 1].assign(cap, 0);
        outs[2 * p] = bufs[2 * p].data(); outs[2 * p + 1] = bufs[2 * p + 1].data();
      }
      // the device aligner takes a bounded batch per call: 64 pairs to a wave, 65 536 pairs fill the 1 024 SIMDs once
      const size_t CH = 65536;
      for (size_t p0 = 0; p0 < P; p0 += CH) {
        const int n = (int)std::min(CH, P - p0);
        const int rc = dada2hip_nwvec(n, s1.data() + p0, s2.data() + p0, match, mismatch, gap, /*band=*/-1, /*endsfree=*/1, device,
                                      outs.data() + 2 * p0, errbuf, errlen);
        if (rc != DADA2HIP_OK) return rc;
      }
      for (size_t p = 0; p < P; p++) {
        const std::string a1(outs[2 * p]), a2(outs[2 * p + 1]);
        eval_pair(a1, a2, nmatch[p], nmismatch[p], nindel[p]);
        prefer[p] = 1 + (n0R[ups[p].second - 1] > n0F[ups[p].first - 1] ? 1 : 0);                      // paired.R:164
        accept[p] = (nmatch[p] >= min_overlap && nmismatch[p] + nindel[p] <= max_mismatch) ? 1 : 0;    // :165
        if (accept[p]) seq[p] = pair_consensus(a1, a2, prefer[p], trim_overhang != 0);                 // :167, rejects get "" (:174)
      }
    }
  }
  // order(abundance, decreasing=TRUE) is stable (paired.R:182)
  std::vector<int32_t> ord(P);
  for (size_t p = 0; p < P; p++) ord[p] = (int32_t)p;
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return abund[a] > abund[b]; });
  for (size_t k = 0; k < P; k++) {
    const int32_t p = ord[k];
    m->sequence.push_back(std::move(seq[p]));
    m->abundance.push_back(abund[p]); m->forward.push_back(ups[p].first); m->reverse.push_back(ups[p].second);
    m->nmatch.push_back(nmatch[p]); m->nmismatch.push_back(nmismatch[p]); m->nindel.push_back(nindel[p]);
    m->prefer.push_back(prefer[p]); m->accept.push_back(accept[p]);
  }
  *out = m.release();
  return DADA2HIP_OK;
}

// no exception crosses the C ABI (include/dada2hip.h)
int dada2hip_merge_pairs(int64_t nreads, const int32_t *fwd, const int32_t *rev, int32_t nF, const char *const *seqsF,
                         const int32_t *n0F, int32_t nR, const char *const *seqsR, const int32_t *n0R, int32_t min_overlap,
                         int32_t max_mismatch, int32_t trim_overhang, int32_t just_concatenate, int32_t device,
                         dada2hip_mergers **out, char *errbuf, size_t errlen) {
  try {
    return merge_pairs_body(nreads, fwd, rev, nF, seqsF, n0F, nR, seqsR, n0R, min_overlap, max_mismatch, trim_overhang,
                            just_concatenate, device, out, errbuf, errlen);
  } catch (const std::bad_alloc &) {
    set_err(errbuf, errlen, "dada2hip: out of host memory in merge_pairs");
  } catch (const std::exception &e) {
    set_err(errbuf, errlen, (std::string("dada2hip: ") + e.what()).c_str());
  } catch (...) {
    set_err(errbuf, errlen, "dada2hip: unknown error in merge_pairs");
  }
  if (out) *out = nullptr;
  return DADA2HIP_ERR_RUNTIME;
}

int32_t dada2hip_mergers_nrow(const dada2hip_mergers *m) { return m ? (int32_t)m->abundance.size() : 0; }
const char *dada2hip_mergers_sequence(const dada2hip_mergers *m, int32_t i) {
  return (m && i >= 0 && (size_t)i < m->sequence.size()) ? m->sequence[i].c_str() : nullptr;
}
const int32_t *dada2hip_mergers_abundance(const dada2hip_mergers *m) { return m ? m->abundance.data() : nullptr; }
const int32_t *dada2hip_mergers_forward(const dada2hip_mergers *m) { return m ? m->forward.data() : nullptr; }
const int32_t *dada2hip_mergers_reverse(const dada2hip_mergers *m) { return m ? m->reverse.data() : nullptr; }
const int32_t *dada2hip_mergers_nmatch(const dada2hip_mergers *m) { return m ? m->nmatch.data() : nullptr; }
const int32_t *dada2hip_mergers_nmismatch(const dada2hip_mergers *m) { return m ? m->nmismatch.data() : nullptr; }
const int32_t *dada2hip_mergers_nindel(const dada2hip_mergers *m) { return m ? m->nindel.data() : nullptr; }
const int32_t *dada2hip_mergers_prefer(const dada2hip_mergers *m) { return m ? m->prefer.data() : nullptr; }
const int32_t *dada2hip_mergers_accept(const dada2hip_mergers *m) { return m ? m->accept.data() : nullptr; }
void dada2hip_mergers_free(dada2hip_mergers *m) { delete m; }

}  // extern "C"
