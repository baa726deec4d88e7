package sw ;

import org.apache.spark.api.java.function.PairFlatMapFunction ;
import org.apache.spark.api.java.function.PairFunction ;

import scala.Tuple2 ;
import scala.Tuple3 ;

import java.nio.ByteBuffer ;
import java.nio.charset.StandardCharsets ;
import java.util.ArrayList ;
import java.util.Iterator ;

/**
 * Drop-in replacements for Distribution.MapRef (src/sw/Distribution.java:383-437) that run the matrix fill,
 * tied-maximum search and traceback of every (reference, read) pair on an MI355X through libswmi.so.
 *
 * NOT compiled by this repository's build (no JDK in the build image); see INTEGRATION.md.
 *
 *   mapRDD = listRDD.mapToPair( new MapRef() ) ;                           // reference, Distribution.java:338
 *   mapRDD = listRDD.mapPartitionsToPair( new GpuSmithWaterman.MapPartition() ) ;   // one native call per partition
 *   mapRDD = listRDD.mapToPair( new GpuSmithWaterman.MapRef() ) ;           // per element (simple, slower)
 */
public class GpuSmithWaterman
{
	static { System.loadLibrary( "swmi_jni" ) ; }

	public static final int TIE_SERIAL = 0 ;	// SmithWaterman.GetCellScore order
	public static final int TIE_STRICT = 1 ;	// DistributedSW.GetCellScore order

	static native long nativeCreate( int device ) ;
	static native void nativeDestroy( long ctx ) ;
	static native long nativeAlignBatch( long ctx , int match , int mismatch , int gap , int tieMode , byte[] types ,
			ByteBuffer refBytes , long[] refOff , int nRefs , ByteBuffer readBytes , long[] readOff , int nReads ) ;
	static native void nativeFreeBatch( long ctx , long batch ) ;
	static native int nativeRefTotal( long batch , int ref ) ;
	static native long nativeRefSiteCount( long batch , int ref ) ;
	static native byte[][] nativeRefSite( long batch , int ref , long k , int[] begin ) ;
	/** {number of match sites, bytes of all their strings} of the references refLo .. refHi-1 */
	static native long[] nativeRefSitesSizes( long batch , int refLo , int refHi ) ;
	/** MapRef's output of the references refLo .. refHi-1 into the caller's arrays, ONE call (swmi_ref_sites_packed, include/swmi.h) */
	static native void nativeRefSitesPacked( long batch , int refLo , int refHi , int[] totals , long[] degenerate , long[] siteFirst ,
			int[] begins , int[] lens , long[] strOff , byte[] blob ) ;

	/** one context per executor thread: MapRef.call runs concurrently on every task thread */
	private static final java.util.concurrent.ConcurrentHashMap<Long,Long> CONTEXTS = new java.util.concurrent.ConcurrentHashMap<Long,Long>() ;
	static
	{
		// executor threads outlive tasks; whatever is still open when the JVM goes down is destroyed here
		Runtime.getRuntime().addShutdownHook( new Thread() { @Override public void run() { releaseAllContexts() ; } } ) ;
	}

	private static long context()
	{
		Long tid = Long.valueOf( Thread.currentThread().getId() ) ;
		Long ctx = CONTEXTS.get( tid ) ;
		if( ctx == null )
		{
			int nGpus = Integer.getInteger( "swmi.gpus" , 8 ) ;
			ctx = Long.valueOf( nativeCreate( (int)( tid.longValue() % nGpus ) ) ) ;
			CONTEXTS.put( tid , ctx ) ;
		}
		return ctx.longValue() ;
	}

	/** destroys the calling thread's context (device buffers, HIP stream): call it when a task thread retires */
	public static void releaseThreadContext()
	{
		Long ctx = CONTEXTS.remove( Long.valueOf( Thread.currentThread().getId() ) ) ;
		if( ctx != null ) nativeDestroy( ctx.longValue() ) ;
	}

	/** destroys every context (no native call may be in flight) */
	public static synchronized void releaseAllContexts()
	{
		for( Long tid : new ArrayList<Long>( CONTEXTS.keySet() ) )
		{
			Long ctx = CONTEXTS.remove( tid ) ;
			if( ctx != null ) nativeDestroy( ctx.longValue() ) ;
		}
	}

	/** a direct ByteBuffer holds at most Integer.MAX_VALUE bytes: one native call takes at most this many sequence bytes */
	private static final long MAX_BYTES_PER_CALL = 1L << 30 ;
	private static final Tuple2<Integer,String[]> EMPTY_SITE = new Tuple2<Integer,String[]>( Integer.valueOf(0) , new String[]{ "" , "" } ) ;

	/**
	 * Characters above U+00FF cannot be narrowed to ISO-8859-1: map Character.toUpperCase(c) of every distinct
	 * such character of the partition to a free byte value first (the aligned strings come back as bytes of the
	 * sequences as uploaded, so the mapping must be undone on the way out).  DNA/IUPAC input never needs it.
	 */
	private static ByteBuffer pack( java.util.List<String> seqs , long[] off )
	{
		long total = 0 ;
		for( int i = 0 ; i < seqs.size() ; i++ ) { off[i] = total ; total += seqs.get(i).length() ; }
		off[seqs.size()] = total ;
		if( total > Integer.MAX_VALUE )		// (MapPartition never hands over more than MAX_BYTES_PER_CALL of references; reads are one list)
			throw new IllegalArgumentException( "more than 2 GiB of sequence in one native call: " + total ) ;
		ByteBuffer buf = ByteBuffer.allocateDirect( (int) Math.max(total,1L) ) ;
		for( String s : seqs ) buf.put( s.getBytes(StandardCharsets.ISO_8859_1) ) ;
		return buf ;
	}

	/** mapPartitionsToPair variant: the elements of the partition in ONE native call per (at most) 1 GiB of references. */
	public static class MapPartition implements PairFlatMapFunction< Iterator<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> , Integer , Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>> >
	{
		@Override
		public Iterable<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>> call( Iterator<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> it )
		{
			ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> elems = new ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>>() ;
			while( it.hasNext() ) elems.add( it.next() ) ;
			ArrayList<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>> out = new ArrayList<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>>( elems.size() ) ;
			int lo = 0 ;
			while( lo < elems.size() )
			{
				// as many elements as fit one native call
				long bytes = 0 ;
				int hi = lo ;
				while( hi < elems.size() && ( hi == lo || bytes + elems.get(hi)._1()[1].length() <= MAX_BYTES_PER_CALL ) ) { bytes += elems.get(hi)._1()[1].length() ; hi++ ; }
				alignRange( elems , lo , hi , out ) ;
				lo = hi ;
			}
			return out ;
		}

		private static void alignRange( ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> elems , int lo , int hi ,
				ArrayList<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>> out )
		{
			// CombineReadsToRef (Distribution.java:714-724) hands every element the same reads and algoArgs
			ArrayList<String> reads = elems.get(lo)._2() ;
			int[] sc = elems.get(lo)._3()._1() ;
			char[] ty = elems.get(lo)._3()._2() ;
			int n = hi - lo ;
			ArrayList<String> refs = new ArrayList<String>( n ) ;
			for( int e = lo ; e < hi ; e++ ) refs.add( elems.get(e)._1()[1] ) ;

			long[] refOff = new long[n+1] , readOff = new long[reads.size()+1] ;
			ByteBuffer refBuf = pack( refs , refOff ) , readBuf = pack( reads , readOff ) ;
			byte[] types = { (byte)ty[0] , (byte)ty[1] , (byte)ty[2] , (byte)ty[3] } ;

			long ctx = context() ;
			long batch = nativeAlignBatch( ctx , sc[0] , sc[1] , sc[2] , TIE_SERIAL , types , refBuf , refOff , n , readBuf , readOff , reads.size() ) ;
			try
			{
				// everything MapRef returns for these references in TWO native calls: the sizes, then the data
				long[] sizes = nativeRefSitesSizes( batch , 0 , n ) ;
				if( sizes[0] > Integer.MAX_VALUE || sizes[1] > Integer.MAX_VALUE )
					throw new IllegalStateException( "the match sites of one native call do not fit Java arrays: " + sizes[0] + " sites, " + sizes[1] + " bytes" ) ;
				int[] totals = new int[n] , begins = new int[(int)sizes[0]] , lens = new int[(int)sizes[0]] ;
				long[] degenerate = new long[n] , siteFirst = new long[n+1] , strOff = new long[(int)sizes[0]] ;
				byte[] blob = new byte[(int)sizes[1]] ;
				nativeRefSitesPacked( batch , 0 , n , totals , degenerate , siteFirst , begins , lens , strOff , blob ) ;
				for( int r = 0 ; r < n ; r++ )
				{
					long nSites = degenerate[r] + siteFirst[r+1] - siteFirst[r] ;
					if( nSites > Integer.MAX_VALUE ) throw new IllegalStateException( "more match sites than an ArrayList holds: " + nSites ) ;
					ArrayList<Tuple2<Integer,String[]>> sites = new ArrayList<Tuple2<Integer,String[]>>( (int)nSites ) ;
					// a pair whose maximum is 0 yields (0, "", "") for every one of its m*n cells (SmithWaterman.java:154,182-185,378-380):
					// begin 0 sorts them in front of every real site; one shared tuple stands for all of them
					for( long d = 0 ; d < degenerate[r] ; d++ ) sites.add( EMPTY_SITE ) ;
					for( int s = (int)siteFirst[r] ; s < (int)siteFirst[r+1] ; s++ )
					{
						int at = (int)strOff[s] ;
						String[] aligned = { new String( blob , at , lens[s] , StandardCharsets.ISO_8859_1 ) ,
								new String( blob , at + lens[s] , lens[s] , StandardCharsets.ISO_8859_1 ) } ;
						sites.add( new Tuple2<Integer,String[]>( Integer.valueOf(begins[s]) , aligned ) ) ;
					}
					out.add( new Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>( Integer.valueOf(totals[r]) ,
							new Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>( elems.get(lo+r)._1() , sites ) ) ) ;
				}
			}
			finally { nativeFreeBatch( ctx , batch ) ; }
		}
	}

	/** Same signature as Distribution.MapRef: per-element drop-in. */
	public static class MapRef implements PairFunction< Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>> , Integer , Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>> >
	{
		@Override
		public Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>> call( Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>> tuple )
		{
			ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> one = new ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>>(1) ;
			one.add( tuple ) ;
			return new MapPartition().call( one.iterator() ).iterator().next() ;
		}
	}
}
