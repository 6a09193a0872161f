// adapter/shim/gen/hip_util.cc.nesi_gen.cc -- stand-in for the file Boda's NESI generator writes for hip_util.cc (class info + factory that
// maps type_id "hip" to hip_compute_t); included inside namespace boda at the end of hip_util.cc, like the generated one.
cinfo_t const * hip_compute_t::get_cinfo( void ) const { return 0; }
p_rtc_compute_t make_rtc_compute_by_type_id( string const & be, uint32_t const device ) {
  if( be != "hip" ) { rt_err( "unknown rtc_compute_t type_id '" + be + "'" ); }
  shared_ptr< hip_compute_t > r = make_shared< hip_compute_t >(); r->be = be; r->device = device; r->gen_src = 0; r->gen_src_output_dir.exp = "rtc-gen-src";
  return r;
}
p_rtc_compute_t make_rtc_compute_by_type_id_devices( string const & be, vect_uint32_t const & devices ) {   // (be=hip,devices=0:1:...) as NESI would fill it
  p_rtc_compute_t r = make_rtc_compute_by_type_id( be, 0 ); dynamic_cast< hip_compute_t & >( *r ).devices = devices; return r;
}
