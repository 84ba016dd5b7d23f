/* -*- c++ -*- */
/* SWIG interface of the gfx950 MUSIC-DoA block: the stanza of gr-baz's swig/baz_swig.i:560-574 with its guard
 * swapped from ARMADILLO_FOUND to BAZ_MUSIC_HIP_FOUND (the block no longer needs Armadillo; it needs the HIP
 * kernel library).  Two ways to use it on a GNU Radio 3.7 host (see INTEGRATION.md):
 *   - inside gr-baz: `%include "baz_music.i"` from baz_swig.i in place of lines 558-576, so that the python
 *     surface stays `baz.music_doa(m, n, nsamples, array_response, resolution)` / `.set_array_response(...)`;
 *   - stand-alone: swig/CMakeLists.txt of this tree wraps it as the module `baz_music_swig` (BAZ_MUSIC_SWIG_MODULE),
 *     which python/__init__ re-exports into the `baz` namespace.
 * The class body names exactly what the reference exposes: the private constructor, the factory and
 * set_array_response (lib/baz_music_doa.h:29-60). */

#ifdef BAZ_MUSIC_SWIG_MODULE
%include "gnuradio.i"			// the common stuff
%include "pycontainer.swg"
%include "config.h"

#undef BAZ_API
#define BAZ_API

%{
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
%}
#endif // BAZ_MUSIC_SWIG_MODULE

#ifdef BAZ_MUSIC_HIP_FOUND

%{
#include "baz_music_doa.h"
%}

GR_SWIG_BLOCK_MAGIC(baz,music_doa)

baz_music_doa_sptr baz_make_music_doa(unsigned int m, unsigned int n, unsigned int nsamples, const /*array_response_t*/std::vector<std::vector<gr_complex> >& array_response, unsigned int resolution);

class baz_music_doa : public gr::sync_block
{
private:
	baz_music_doa(unsigned int m, unsigned int n, unsigned int nsamples, const array_response_t& array_response, unsigned int resolution);
public:
	void set_array_response(const /*array_response_t*/std::vector<std::vector<gr_complex> >& array_response);
	// opt-in extension (not in the reference; default off = reference behaviour): n strongest local maxima
	void set_peak_mode(bool on);
};

#endif // BAZ_MUSIC_HIP_FOUND
